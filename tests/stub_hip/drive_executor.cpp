// Drives the HOST side of the native step executor (vn_net_prepare x2 -> vn_net_forward -> vn_net_backward: the calls
// voxelnet_amd/model.py makes per train step, train.py:148-151) with fake device pointers, under the stub HIP layer of
// stub_hip.c.  Prints "<launches per step> <microseconds per step>" (median of 5 blocks).  Test infrastructure.
//   drive_executor <libvoxelnet_hip.so> <steps> [bucket_events [mode [H W [K]]]]   (mode 0 bf16, 1 fp32, 2 fp32x3)
// bucket_events = 2: the whole train step as ONE call (vn_net_step: + voxel feature encoder, loss, clip + SGD); -1: control.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <vector>
#include "../../include/voxelnet_hip.h"

extern "C" unsigned long long vn_stub_launches(void);

#define SYM(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(lib, #name)); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    const int steps = atoi(argv[2]);
    const int buckets = argc > 3 ? atoi(argv[3]) : 0;
    if (buckets < 0) {
        // control: a private-memory workload of the executor's kind (struct fills and copies over ~48 KB, some integer
        // arithmetic) that shares NOTHING between processes — what eight copies of it lose against one is the machine's
        // doing (SMT siblings, shared caches, a noisy host), the yardstick for the executor's own ratio
        std::vector<char> a(48 << 10), b(48 << 10);
        std::vector<double> blocks;
        unsigned acc = 1;
        for (int blk = 0; blk < 5; ++blk) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < steps; ++i) {
                memset(a.data(), i & 0xff, a.size());
                memcpy(b.data(), a.data(), a.size());
                for (size_t k = 0; k < b.size(); k += 64) acc = acc * 1664525u + (unsigned char)b[k];
            }
            blocks.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps);
        }
        std::sort(blocks.begin(), blocks.end());
        printf("%u %.2f\n", acc & 1u, blocks[2]);
        return 0;
    }
    SYM(vn_net_workspace_bytes) SYM(vn_net_create) SYM(vn_net_destroy) SYM(vn_net_prepare) SYM(vn_net_forward) SYM(vn_net_backward)
    SYM(vn_net_step) SYM(vn_vfe_workspace_bytes) SYM(vn_rpn_loss_workspace_bytes) SYM(vn_clip_sgd_workspace_bytes)
    vnNetConfig cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.B = 2; cfg.D = 10; cfg.H = 400; cfg.W = 352; cfg.block1_stride = 2; cfg.mode = 0; cfg.training = 1; cfg.sparse_first = 1;
    int64_t K = 12345;
    if (argc > 4) cfg.mode = atoi(argv[4]);
    if (argc > 6) { cfg.H = atoi(argv[5]); cfg.W = atoi(argv[6]); }
    if (argc > 7) K = atoll(argv[7]);
    const size_t ws_bytes = p_vn_net_workspace_bytes(&cfg, K);
    if (!ws_bytes) return 3;
    // fake device memory: addresses only (the stub never dereferences; the executor's host code never does either)
    char *const dev = reinterpret_cast<char *>(0x7f0000000000ull);
    char *ws = dev;
    float *par = reinterpret_cast<float *>(dev + (ws_bytes + 4096) / 256 * 256);
    vnLayerParams L[23];
    vnLayerGrads G[23];
    for (int l = 0; l < 23; ++l) {
        L[l] = vnLayerParams{par, par + 1024, par + 2048, par + 3072, par + 4096, par + 5120};
        G[l] = vnLayerGrads{par + 6144, par + 7168, par + 8192, par + 9216};
        par += 1 << 20;
    }
    float *heads_w = par, *heads_b = par + 16384, *prob = par + 32768, *reg = par + (1 << 22), *dprob = par + (2 << 22),
          *dreg = par + (3 << 22), *dhw = par + (4 << 22), *dhb = par + (5 << 22), *d_in = par + (6 << 22);
    const int64_t *coord = reinterpret_cast<const int64_t *>(par + (7 << 22));
    const void *vw_rows = par + (8 << 22);
    vnNet *net = nullptr;
    if (p_vn_net_create(&net)) return 4;
    vnStream main_s = reinterpret_cast<vnStream>(0x10), side_s = reinterpret_cast<vnStream>(0x20);
    vnStep st;
    memset(&st, 0, sizeof(st));
    {
        float *q = par + (10 << 22);
        auto take = [&](size_t floats) { float *r = q; q += (floats + 63) / 64 * 64; return r; };
        st.feature = take(K * 35 * 7); st.coord = coord; st.K = K; st.T = 35; st.bn_momentum = 0.1f; st.bn_eps = 1e-5f;
        st.vfe = vnVfeWeights{take(112), take(16), take(16), take(16), take(16), take(16), take(2048), take(64), take(64), take(64), take(64), take(64)};
        st.vfe_grads = vnVfeGrads{take(112), take(16), take(16), take(16), take(2048), take(64), take(64), take(64)};
        st.vfe_ws_bytes = p_vn_vfe_workspace_bytes(K, 35); st.vfe_ws = take(st.vfe_ws_bytes / 4 + 1);
        st.voxelwise = take(K * 128); st.vfe_stats = take(320);
        st.vw_rows = cfg.mode == 0 ? (void *)take(K * 64) : (void *)st.voxelwise;
        st.d_voxelwise = take(K * 128);
        st.prob_w = take(2 * 768); st.prob_b = take(2); st.reg_w = take(14 * 768); st.reg_b = take(14);
        st.heads_w = heads_w; st.heads_b = heads_b; st.d_heads_w = dhw; st.d_heads_b = dhb;
        st.layers = L; st.grads = G; st.ws = ws; st.ws_bytes = ws_bytes;
        st.prob = prob; st.reg = reg; st.d_prob = dprob; st.d_reg = dreg;
        const int hf = cfg.H / cfg.block1_stride, wf = cfg.W / cfg.block1_stride;
        st.pos = take((size_t)cfg.B * hf * wf * 2); st.neg = take((size_t)cfg.B * hf * wf * 2); st.targets = take((size_t)cfg.B * hf * wf * 14);
        st.targets_stream = reinterpret_cast<vnStream>(0x30);
        st.alpha = 1.5f; st.beta = 1.f; st.sigma = 3.f;
        st.loss_ws_bytes = p_vn_rpn_loss_workspace_bytes(cfg.B, hf, wf); st.loss_ws = take(st.loss_ws_bytes / 4 + 1);
        st.loss5 = take(5); st.g_loss = take(1);
        st.n_chunks = 2000; st.chunks = reinterpret_cast<const vnParamChunk *>(take(2000 * sizeof(vnParamChunk) / 4));
        st.max_norm = 5.f; st.lr = 0.01f; st.opt_ws_bytes = p_vn_clip_sgd_workspace_bytes(2000); st.opt_ws = take(st.opt_ws_bytes / 4 + 1);
        st.total_norm = take(1);
        st.bn_counters = reinterpret_cast<int64_t *const *>(take(64)); st.n_bn_counters = 25;
        st.stream = reinterpret_cast<vnStream>(0x10); st.side_stream = reinterpret_cast<vnStream>(0x20);
    }
    auto one_step = [&]() -> int {
        int rc;
        if (buckets == 2) {
            cfg.bucket_events = 0; cfg.defer_join = 0; cfg.prepared = 0;
            return p_vn_net_step(net, &cfg, &st);
        }
        cfg.bucket_events = 0; cfg.defer_join = 0;
        cfg.prepared = 1;
        if ((rc = p_vn_net_prepare(net, &cfg, L, nullptr, coord, K, ws, ws_bytes, side_s))) return rc;
        cfg.prepared = 2;
        if ((rc = p_vn_net_prepare(net, &cfg, L, heads_w, coord, K, ws, ws_bytes, side_s))) return rc;
        cfg.prepared = 1;
        if ((rc = p_vn_net_forward(net, &cfg, L, heads_w, heads_b, nullptr, coord, vw_rows, K, ws, ws_bytes, prob, reg, main_s, side_s))) return rc;
        cfg.bucket_events = buckets; cfg.defer_join = 1;
        return p_vn_net_backward(net, &cfg, L, heads_w, dprob, dreg, prob, nullptr, coord, vw_rows, K, ws, ws_bytes, G, dhw, dhb, d_in,
                                 0, 24, main_s, side_s);
    };
    for (int i = 0; i < 20; ++i)
        if (int rc = one_step()) { fprintf(stderr, "step failed: %d\n", rc); return 5; }
    std::vector<double> blocks;
    unsigned long long l0 = vn_stub_launches();
    for (int b = 0; b < 5; ++b) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i)
            if (int rc = one_step()) { fprintf(stderr, "step failed: %d\n", rc); return 5; }
        blocks.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps);
    }
    const double per_step = (double)(vn_stub_launches() - l0) / (5.0 * steps);
    std::sort(blocks.begin(), blocks.end());
    printf("%.1f %.2f\n", per_step, blocks[2]);
    p_vn_net_destroy(net);
    return 0;
}
