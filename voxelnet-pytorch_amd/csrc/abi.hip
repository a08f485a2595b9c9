#include "common.h"
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" int vn_abi_version(void) { return 3; }

namespace {
// every tuning aid of the library (vn_knob): kernel-selection overrides used for A/B measurements (tools/README.md)
const char *const KNOBS[] = {"VN_BN_FUSE_ROWS", "VN_BN_HOIST", "VN_BOX_SIDE", "VN_BOX_ZERO", "VN_DUP", "VN_EARLY_DECONV", "VN_EARLY_UNPACK", "VN_FUSE_BWD_REDUCE", "VN_GG_CONFIG", "VN_GG_ROWS_CONFIG", "VN_HEADS_BLOCKS", "VN_HEADS_NOCAT", "VN_HEADS_STREAM", "VN_M0_BN", "VN_M0_MAIN", "VN_P2D_ROT", "VN_PATCH", "VN_PATCH2D", "VN_PATCH2D_WAVES", "VN_PATCH_P0", "VN_PATCH_P1", "VN_SKIP", "VN_UNPACK_M2", "VN_WGP2_BLOCKS", "VN_WGP2_STAGES", "VN_WGP_BLOCKS", "VN_WGRAD_PATCH", "VN_WG_BLOCKS", "VN_WG_EARLY", "VN_WG_STREAMS", "VN_WG_TRI_WAVES", "VN_WG_WAVES", "VN_WG_XCD", "VN_X3_PRESPLIT"};
constexpr int NKNOBS = sizeof(KNOBS) / sizeof(KNOBS[0]);
}  // namespace

int vn_knob(const char *name, int dflt) {
    for (int i = 0; i < NKNOBS; ++i)
        if (!strcmp(KNOBS[i], name)) {
            const char *e = getenv(name);
            return e && *e ? atoi(e) : dflt;
        }
    // a name missing from the table would make the knob a silent no-op: say so, once per name (stderr; no abort: this is a
    // tuning aid of a library, and the default is what the product runs)
    static std::mutex mu;
    static const char *seen[16];
    static int nseen = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < nseen; ++i)
        if (!strcmp(seen[i], name)) return dflt;
    if (nseen < 16) seen[nseen++] = name;
    fprintf(stderr, "libvoxelnet_hip: vn_knob(\"%s\") is not in the KNOBS table of csrc/abi.hip — the default %d is used and the "
                    "environment is NOT read\n", name, dflt);
    return dflt;
}

// "libvoxelnet_hip gfx950 (CDNA4) <date> <time>" + " overrides: NAME=value ..." for every tuning variable that is set in the
// environment (built once, at the first call: the knobs themselves are read once per process too)
extern "C" const char *vn_build_info(void) {
    static char buf[1536];
    static std::once_flag once;
    std::call_once(once, [] {
        size_t n = (size_t)snprintf(buf, sizeof(buf), "libvoxelnet_hip gfx950 (CDNA4) abi %d " __DATE__ " " __TIME__, vn_abi_version());
        bool any = false;
        for (int i = 0; i < NKNOBS && n + 64 < sizeof(buf); ++i) {
            const char *e = getenv(KNOBS[i]);
            if (!e || !*e) continue;
            n += (size_t)snprintf(buf + n, sizeof(buf) - n, "%s%s=%.24s", any ? " " : " overrides: ", KNOBS[i], e);
            any = true;
        }
    });
    return buf;
}
