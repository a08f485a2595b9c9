// Gradient all-reduce of one flat bucket over RCCL (xGMI) — SURVEY.md §8(b) `vn_allreduce_bucket`, §8(e).
// The reference has no distributed code (train.py trains on one device); data parallelism over the GPUs of a node is
// this build's: one process per GPU, each trains on its own point clouds, the 6,809,392 gradient elements are averaged
// once per step in four flat buckets launched while the backward still runs (voxelnet_amd/parallel.py).
//
// RCCL is bound at RUN time (dlopen): the library has no link-time dependency on it, single-GPU users never load it,
// and inside a PyTorch process the already-loaded librccl of torch is reused (one RCCL instance per process).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

struct NcclId { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef void *NcclComm;                          // ncclComm_t
typedef int (*fn_get_id)(NcclId *);
typedef int (*fn_init_rank)(NcclComm *, int, NcclId, int);
typedef int (*fn_destroy)(NcclComm);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_version)(int *);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h (NCCL 2.x)
// The four entry points are declared by hand (no link-time dependency), so the hand-written ABI above — the 128-byte id
// passed BY VALUE, ncclFloat32 = 7, ncclSum = 0 — is only trusted for the major version it was written against:
// ncclGetVersion() must report 2.x (version code = major * 10000 + minor * 100 + patch for >= 2.9, major * 1000 + ... before)
constexpr int NCCL_MAJOR_EXPECTED = 2;

struct Rccl {
    fn_get_id get_id;
    fn_init_rank init_rank;
    fn_destroy destroy;
    fn_allreduce allreduce;
    int version;     // ncclGetVersion code, 0 if unavailable
    bool ok;
};

// immutable after the first call (kernel-selection-table rule of the ABI: no mutable state afterwards)
const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl t{};
        void *h = nullptr;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;      // already in the process (torch's)
        if (!h)
            for (const char *n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return t;
        t.get_id = reinterpret_cast<fn_get_id>(dlsym(h, "ncclGetUniqueId"));
        t.init_rank = reinterpret_cast<fn_init_rank>(dlsym(h, "ncclCommInitRank"));
        t.destroy = reinterpret_cast<fn_destroy>(dlsym(h, "ncclCommDestroy"));
        t.allreduce = reinterpret_cast<fn_allreduce>(dlsym(h, "ncclAllReduce"));
        fn_version ver = reinterpret_cast<fn_version>(dlsym(h, "ncclGetVersion"));
        int code = 0;
        if (ver && ver(&code) == 0) t.version = code;
        const int major = t.version >= 20000 ? t.version / 10000 : t.version / 1000;
        t.ok = t.get_id && t.init_rank && t.destroy && t.allreduce && major == NCCL_MAJOR_EXPECTED;
        return t;
    }();
    return r;
}

__global__ void __launch_bounds__(256) k_scale(float *__restrict__ x, int64_t n4, int64_t n, float s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 v = reinterpret_cast<float4 *>(x)[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        reinterpret_cast<float4 *>(x)[i] = v;
    }
    if (i == 0)
        for (int64_t j = n4 * 4; j < n; ++j) x[j] *= s;
}

}  // namespace

// status: VN_OK, VN_EINVAL, VN_EUNSUPPORTED (no librccl in the process / on the box, or one whose ncclGetVersion is not
// 2.x), or 1000 + ncclResult_t.
// EXPERIMENTAL until it has run on 2+ GPUs (this pool hands out one GPU per call; the default data-parallel path is
// torch.distributed's process group, voxelnet_amd/parallel.py).  A communicator made here lives NEXT TO torch's NCCL
// process group: collectives of the two must never be in flight together in different orders on different ranks —
// GradAllReducer issues its buckets in one fixed order on every rank and the caller must not run a torch.distributed
// collective between launch_bucket(0) and finish().

// RCCL version code the library bound (ncclGetVersion), 0 when no usable librccl was found
extern "C" int vn_comm_rccl_version(void) { return rccl().version; }

extern "C" int vn_comm_unique_id(void *id128) {
    VN_CHECK_ARG(id128);
    const Rccl &r = rccl();
    if (!r.ok) return VN_EUNSUPPORTED;
    const int e = r.get_id(static_cast<NcclId *>(id128));
    return e ? 1000 + e : VN_OK;
}

extern "C" int vn_comm_create(void **nccl_comm, const void *id128, int32_t world, int32_t rank) {
    VN_CHECK_ARG(nccl_comm && id128 && world >= 1 && rank >= 0 && rank < world);
    const Rccl &r = rccl();
    if (!r.ok) return VN_EUNSUPPORTED;
    NcclId id;
    memcpy(&id, id128, sizeof(id));
    NcclComm c = nullptr;
    const int e = r.init_rank(&c, world, id, rank);     // collective: every rank calls it, on its own current device
    if (e) return 1000 + e;
    *nccl_comm = c;
    return VN_OK;
}

extern "C" int vn_comm_destroy(void *nccl_comm) {
    if (!nccl_comm) return VN_OK;
    const Rccl &r = rccl();
    if (!r.ok) return VN_EUNSUPPORTED;
    const int e = r.destroy(nccl_comm);
    return e ? 1000 + e : VN_OK;
}

// bucket[i] = sum over ranks of (scale * bucket[i]), in place, on `stream` (asynchronous).  scale = 1 / world gives the
// mean of stock DDP; scaling BEFORE the sum keeps every rank's result bit-identical (the same ring order everywhere).
extern "C" int vn_allreduce_bucket(void *nccl_comm, float *bucket, int64_t count, float scale, vnStream stream) {
    VN_CHECK_ARG(nccl_comm && count >= 0 && (bucket || count == 0));
    VN_CHECK_ARG((reinterpret_cast<uintptr_t>(bucket) & 15) == 0);
    if (count == 0) return VN_OK;
    const Rccl &r = rccl();
    if (!r.ok) return VN_EUNSUPPORTED;
    hipStream_t st = vn_stream(stream);
    if (scale != 1.0f) {
        const int64_t n4 = count / 4;
        const int64_t blocks = vn_ceil_div(n4 > 0 ? n4 : 1, 256);
        k_scale<<<(unsigned)blocks, 256, 0, st>>>(bucket, n4, count, scale);
        VN_LAUNCH_STATUS();
    }
    const int e = r.allreduce(bucket, bucket, (size_t)count, NCCL_FLOAT32, NCCL_SUM, nccl_comm, st);
    return e ? 1000 + e : VN_OK;
}
