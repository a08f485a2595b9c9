// RPN training targets on the device — utils.generate_targets (utils.py:376-473) with bbox_iou (utils.py:344-373) and
// anchor_to_standup_box2d (utils.py:213-227) folded in: 70,400 anchors x G ground-truth boxes per sample, which the
// reference walks in Python loops on the CPU in every step (model.py:309) before six host-to-device copies.
// The host side (voxelnet_amd/targets.py) parses the KITTI label lines into lidar boxes and their stand-up
// rectangles (O(G) NumPy work, as utils.label_to_gt_box_3d / center_to_corner_box_2d do); everything per anchor is here.
//
// Reference semantics kept exactly (integer results bit-exact against the oracle):
//   * anchor stand-up boxes are the zero-extent (x1,y1,x1,y1) the reference computes (utils.py:219-225);
//   * IoU in float32 with the "+1" conventions and the (y1-x1+1)*(y2-y1+1) union term (utils.py:367);
//   * positives: iou > pos_iou (float32 compare), plus, per box, the FIRST anchor of maximal IoU when that IoU > 0;
//     an anchor's box is the first box with iou > pos_iou, else the first box it is the best anchor of (np.unique's
//     first occurrence in the concatenated list, utils.py:423-427);
//   * negatives: iou < neg_iou for every box (all anchors when there is no box); an anchor may be both;
//   * regression targets in float64, stored as float32 (model.py:329).
// Three launches: per-workgroup argmax partials (value, first index) -> per-box argmax (fixed order) -> per-anchor
// classification and encoding (the IoUs are recomputed: G*12 flops per anchor, cheaper than storing them).
// HBM-bound and tiny: 56 B read + 36 B written per anchor.
#include "common.h"

namespace {

constexpr int TG_THREADS = 256;
constexpr int TG_MAX_GT = VN_TARGETS_MAX_GT;

struct Box2 { float x1, y1, x2, y2; };

// utils.py:344-373 for one (anchor, box) pair; float32 arithmetic in the reference's operation order
__device__ __forceinline__ float iou_pair(const Box2 a, const Box2 g) {
    const float iw = (fminf(a.x2, g.x2) - fmaxf(a.x1, g.x1)) + 1.0f;
    if (!(iw > 0.f)) return 0.f;
    const float ih = (fminf(a.y2, g.y2) - fmaxf(a.y1, g.y1)) + 1.0f;
    if (!(ih > 0.f)) return 0.f;
    const float area = ((g.x2 - g.x1) + 1.0f) * ((g.y2 - g.y1) + 1.0f);
    const float ua = (((a.y1 - a.x1) + 1.0f) * ((a.y2 - a.y1) + 1.0f) + area) - iw * ih;
    return (iw * ih) / ua;      // may be +-inf when the quirky union is exactly 0, as in the reference
}

// utils.py:213-227 on (x, y, w, l): both corners at (x - l/2, y - w/2) for the 0-degree anchor, (x - w/2, y - l/2) for
// the 90-degree one; float64 arithmetic, float32 storage (utils.py:400)
__device__ __forceinline__ Box2 anchor_standup(const double *__restrict__ a, int n) {
    const double x = a[0], y = a[1], w = a[4], l = a[5];
    const float x1 = (float)((n & 1) ? x - w / 2 : x - l / 2);
    const float y1 = (float)((n & 1) ? y - l / 2 : y - w / 2);
    return Box2{x1, y1, x1, y1};
}

// pass 1: part[(b*nblk + blk)*max_gt + k] = (max IoU over this workgroup's anchors, its first index)
__global__ void __launch_bounds__(TG_THREADS) k_tg_argmax_part(const double *__restrict__ anchors, int N,
                                                               const float *__restrict__ gt2d,
                                                               const int32_t *__restrict__ counts, int max_gt,
                                                               float *__restrict__ part_val, int32_t *__restrict__ part_idx) {
    __shared__ Box2 gbox[TG_MAX_GT];
    __shared__ float rv[TG_THREADS / 64];
    __shared__ int ri[TG_THREADS / 64];
    const int b = blockIdx.y, G = counts[b] < max_gt ? counts[b] : max_gt;
    for (int k = threadIdx.x; k < G; k += TG_THREADS) {
        const float *g = gt2d + ((size_t)b * max_gt + k) * 4;
        gbox[k] = Box2{g[0], g[1], g[2], g[3]};
    }
    __syncthreads();
    const int n = blockIdx.x * TG_THREADS + threadIdx.x;
    Box2 a{};
    if (n < N) a = anchor_standup(anchors + (size_t)n * 7, n);
    for (int k = 0; k < G; ++k) {
        float v = n < N ? iou_pair(a, gbox[k]) : -INFINITY;
        int i = n < N ? n : 0x7fffffff;
        // wave argmax: larger value wins, equal values keep the smaller index (np.argmax: first occurrence)
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(i, o, 64);
            if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = v; ri[threadIdx.x >> 6] = i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < TG_THREADS / 64; ++w)
                if (rv[w] > v || (rv[w] == v && ri[w] < i)) { v = rv[w]; i = ri[w]; }
            const size_t o = ((size_t)b * gridDim.x + blockIdx.x) * max_gt + k;
            part_val[o] = v;
            part_idx[o] = i;
        }
        __syncthreads();
    }
}

// pass 2: idmax[b*max_gt + k] = first anchor of maximal IoU with box k, or -1 when that IoU is not > 0 (utils.py:411-415)
__global__ void __launch_bounds__(TG_THREADS) k_tg_argmax_final(const float *__restrict__ part_val,
                                                                const int32_t *__restrict__ part_idx, int nblk,
                                                                const int32_t *__restrict__ counts, int max_gt,
                                                                int32_t *__restrict__ idmax) {
    const int b = blockIdx.x, G = counts[b] < max_gt ? counts[b] : max_gt;
    for (int k = threadIdx.x; k < max_gt; k += TG_THREADS) {
        int best = -1;
        if (k < G) {
            float v = -INFINITY;
            int i = 0x7fffffff;
            for (int blk = 0; blk < nblk; ++blk) {
                const size_t o = ((size_t)b * nblk + blk) * max_gt + k;
                const float ov = part_val[o];
                const int oi = part_idx[o];
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            best = v > 0.f ? i : -1;
        }
        idmax[(size_t)b * max_gt + k] = best;
    }
}

// pass 3: pos / neg flags and the 7 regression targets of every anchor
__global__ void __launch_bounds__(TG_THREADS) k_tg_encode(const double *__restrict__ anchors, int N,
                                                          const double *__restrict__ gt, const float *__restrict__ gt2d,
                                                          const int32_t *__restrict__ counts, int max_gt,
                                                          const int32_t *__restrict__ idmax, float pos_iou, float neg_iou,
                                                          double anchor_h, float *__restrict__ pos, float *__restrict__ neg,
                                                          float *__restrict__ targets) {
    __shared__ Box2 gbox[TG_MAX_GT];
    __shared__ int gmax[TG_MAX_GT];
    const int b = blockIdx.y, G = counts[b] < max_gt ? counts[b] : max_gt;
    for (int k = threadIdx.x; k < G; k += TG_THREADS) {
        const float *g = gt2d + ((size_t)b * max_gt + k) * 4;
        gbox[k] = Box2{g[0], g[1], g[2], g[3]};
        gmax[k] = idmax[(size_t)b * max_gt + k];
    }
    __syncthreads();
    const int n = blockIdx.x * TG_THREADS + threadIdx.x;
    if (n >= N) return;
    const double *a = anchors + (size_t)n * 7;
    const Box2 ab = anchor_standup(a, n);
    int box = -1;
    bool all_neg = true;
    for (int k = 0; k < G; ++k) {
        const float v = iou_pair(ab, gbox[k]);
        if (box < 0 && v > pos_iou) box = k;
        all_neg = all_neg && (v < neg_iou);
    }
    if (box < 0)
        for (int k = 0; k < G; ++k)
            if (gmax[k] == n) { box = k; break; }
    const size_t o = (size_t)b * N + n;
    pos[o] = box >= 0 ? 1.f : 0.f;
    neg[o] = all_neg ? 1.f : 0.f;
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (box >= 0) {
        const double *g = gt + ((size_t)b * max_gt + box) * 7;
        const double diag = sqrt(a[4] * a[4] + a[5] * a[5]);      // utils.py:390-392
        t[0] = (float)((g[0] - a[0]) / diag);
        t[1] = (float)((g[1] - a[1]) / diag);
        t[2] = (float)((g[2] - a[2]) / anchor_h);
        t[3] = (float)log(g[3] / a[3]);
        t[4] = (float)log(g[4] / a[4]);
        t[5] = (float)log(g[5] / a[5]);
        t[6] = (float)(g[6] - a[6]);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) targets[o * 7 + j] = t[j];
}

inline bool tg_args_ok(int32_t B, int32_t N, int32_t max_gt) {
    return B > 0 && N > 0 && max_gt > 0 && max_gt <= TG_MAX_GT && (int64_t)B * N < (1ll << 31) / 8;
}
inline int tg_blocks(int32_t N) { return (N + TG_THREADS - 1) / TG_THREADS; }

}  // namespace

extern "C" size_t vn_rpn_targets_workspace_bytes(int32_t B, int32_t n_anchors, int32_t max_gt) {
    if (!tg_args_ok(B, n_anchors, max_gt)) return 0;
    const size_t part = (size_t)B * tg_blocks(n_anchors) * max_gt;
    return vn_align(part * sizeof(float)) + vn_align(part * sizeof(int32_t)) + vn_align((size_t)B * max_gt * sizeof(int32_t));
}

extern "C" int vn_rpn_targets(const double *anchors, int32_t n_anchors, const double *gt, const float *gt_standup,
                              const int32_t *gt_count, int32_t B, int32_t max_gt, float pos_iou, float neg_iou,
                              double anchor_h, float *pos, float *neg, float *targets, void *workspace,
                              size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(anchors && gt && gt_standup && gt_count && pos && neg && targets && workspace &&
                 tg_args_ok(B, n_anchors, max_gt) && anchor_h != 0.0);
    if (workspace_bytes < vn_rpn_targets_workspace_bytes(B, n_anchors, max_gt)) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    const int nblk = tg_blocks(n_anchors);
    const size_t part = (size_t)B * nblk * max_gt;
    char *ws = static_cast<char *>(workspace);
    float *part_val = reinterpret_cast<float *>(ws);
    int32_t *part_idx = reinterpret_cast<int32_t *>(ws + vn_align(part * sizeof(float)));
    int32_t *idmax = reinterpret_cast<int32_t *>(ws + vn_align(part * sizeof(float)) + vn_align(part * sizeof(int32_t)));
    const dim3 grid((unsigned)nblk, (unsigned)B);
    k_tg_argmax_part<<<grid, TG_THREADS, 0, st>>>(anchors, n_anchors, gt_standup, gt_count, max_gt, part_val, part_idx);
    VN_LAUNCH_STATUS();
    k_tg_argmax_final<<<B, TG_THREADS, 0, st>>>(part_val, part_idx, nblk, gt_count, max_gt, idmax);
    VN_LAUNCH_STATUS();
    k_tg_encode<<<grid, TG_THREADS, 0, st>>>(anchors, n_anchors, gt, gt_standup, gt_count, max_gt, idmax, pos_iou, neg_iou,
                                            anchor_h, pos, neg, targets);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
