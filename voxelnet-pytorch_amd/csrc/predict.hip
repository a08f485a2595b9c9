// Inference tail of RPN3D.predict (model.py:364-395) on the device: deltas_to_boxes_3d (utils.py:476-489), the score
// filter + stand-up rectangles + NMS of filter_boxes (model.py:28-57, utils.nms utils.py:492-553).
// The reference copies both maps to the host, decodes all 70,400 boxes with NumPy and runs the NMS as a Python loop of
// ~10 torch ops per kept box.  Here: one pass over the score map collects the candidates >= SCORE_THRES, and one
// workgroup per sample picks the NMS_POST_TOPK best (only those enter the reference's NMS, utils.py:510), decodes just
// them, and runs the greedy suppression.
// Reference behaviour kept: "box j" = flat elements 7j..7j+6 of the NCHW delta map scored by flat element j of the
// NCHW probability map (reshape without permute, utils.py:478 / model.py:384) on anchor j; float32 boxes from float64
// arithmetic with a float32 exp; float32 corners, float64 stand-up rectangles; areas without "+1"; `IoU <= overlap`
// keeps (a NaN IoU suppresses).  Equal scores: the larger flat index first (the reference's unstable sort leaves
// that order open).
#include "common.h"

namespace {

constexpr int PR_THREADS = 256;
constexpr int PR_MAX_TOPK = VN_PREDICT_MAX_TOPK;

__global__ void __launch_bounds__(PR_THREADS) k_pr_filter(const float *__restrict__ probs, int N, float thres,
                                                          int32_t *__restrict__ count, float *__restrict__ cand_score,
                                                          int32_t *__restrict__ cand_idx) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * PR_THREADS + threadIdx.x;
    if (j >= N) return;
    const float p = probs[(size_t)b * N + j];
    if (p >= thres) {                                   // model.py:34
        const int slot = atomicAdd(count + b, 1);        // order irrelevant: the keys (score, j) are unique
        cand_score[(size_t)b * N + slot] = p;
        cand_idx[(size_t)b * N + slot] = j;
    }
}

// lexicographic key (score, flat index)
__device__ __forceinline__ bool key_less(float s1, int i1, float s2, int i2) { return s1 < s2 || (s1 == s2 && i1 < i2); }

__global__ void __launch_bounds__(PR_THREADS) k_pr_select_nms(const float *__restrict__ deltas, const double *__restrict__ anchors,
                                                              int N, const int32_t *__restrict__ count,
                                                              const float *__restrict__ cand_score,
                                                              const int32_t *__restrict__ cand_idx, int top_k,
                                                              double nms_thres, double anchor_h, float *__restrict__ boxes_out,
                                                              float *__restrict__ scores_out, int32_t *__restrict__ count_out) {
    __shared__ float sel_s[PR_MAX_TOPK];
    __shared__ int sel_i[PR_MAX_TOPK];
    __shared__ float box[PR_MAX_TOPK][7];
    __shared__ double rect[PR_MAX_TOPK][4];
    __shared__ float rs[PR_THREADS / 64];
    __shared__ int ri[PR_THREADS / 64];
    const int b = blockIdx.x;
    const int M = count[b];
    const float *cs = cand_score + (size_t)b * N;
    const int32_t *ci = cand_idx + (size_t)b * N;
    const int n_sel = M < top_k ? M : top_k;
    // ---- the top_k largest keys, in descending order: round t takes the largest key below the previous one
    float prev_s = INFINITY;
    int prev_i = 0x7fffffff;
    for (int t = 0; t < n_sel; ++t) {
        float bs = -INFINITY;
        int bi = -1;
        for (int c = threadIdx.x; c < M; c += PR_THREADS) {
            const float s = cs[c];
            const int i = ci[c];
            if (key_less(s, i, prev_s, prev_i) && key_less(bs, bi, s, i)) { bs = s; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (key_less(bs, bi, os, oi)) { bs = os; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = bs; ri[threadIdx.x >> 6] = bi; }
        __syncthreads();
        for (int w = 0; w < PR_THREADS / 64; ++w)
            if (key_less(bs, bi, rs[w], ri[w])) { bs = rs[w]; bi = ri[w]; }
        if (threadIdx.x == 0) { sel_s[t] = bs; sel_i[t] = bi; }
        prev_s = bs;
        prev_i = bi;
        __syncthreads();
    }
    // ---- decode the selected boxes (utils.py:476-489) and their stand-up rectangles (utils.py:230-252, 283-330)
    if ((int)threadIdx.x < n_sel) {
        const int j = sel_i[threadIdx.x];
        const float *d = deltas + ((size_t)b * N + j) * 7;
        const double *a = anchors + (size_t)j * 7;
        const double diag = sqrt(a[4] * a[4] + a[5] * a[5]);
        float *o = box[threadIdx.x];
        o[0] = (float)((double)d[0] * diag + a[0]);
        o[1] = (float)((double)d[1] * diag + a[1]);
        o[2] = (float)((double)d[2] * anchor_h + a[2]);
        o[3] = (float)((double)expf(d[3]) * a[3]);
        o[4] = (float)((double)expf(d[4]) * a[4]);
        o[5] = (float)((double)expf(d[5]) * a[5]);
        o[6] = (float)((double)d[6] + a[6]);
        const double x = o[0], y = o[1], w = o[4], l = o[5], yaw = o[6];
        const double c = cos(yaw), s = sin(yaw);
        const double fx[4] = {-l / 2, -l / 2, l / 2, l / 2}, fy[4] = {w / 2, -w / 2, -w / 2, w / 2};
        float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float cx = (float)((c * fx[q] + (-s) * fy[q]) + x);      // np.dot row, then + translation; float32 store
            const float cy = (float)((s * fx[q] + c * fy[q]) + y);
            x1 = fminf(x1, cx); x2 = fmaxf(x2, cx);
            y1 = fminf(y1, cy); y2 = fmaxf(y2, cy);
        }
        rect[threadIdx.x][0] = x1; rect[threadIdx.x][1] = y1; rect[threadIdx.x][2] = x2; rect[threadIdx.x][3] = y2;
    }
    __syncthreads();
    // ---- greedy NMS over the <= top_k boxes in descending key order (utils.py:519-551)
    if (threadIdx.x == 0) {
        bool dead[PR_MAX_TOPK];
        for (int t = 0; t < n_sel; ++t) dead[t] = false;
        int kept = 0;
        for (int t = 0; t < n_sel; ++t) {
            if (dead[t]) continue;
            float *ob = boxes_out + ((size_t)b * top_k + kept) * 7;
            for (int q = 0; q < 7; ++q) ob[q] = box[t][q];
            scores_out[(size_t)b * top_k + kept] = sel_s[t];
            ++kept;
            const double ai = (rect[t][2] - rect[t][0]) * (rect[t][3] - rect[t][1]);
            for (int u = t + 1; u < n_sel; ++u) {
                if (dead[u]) continue;
                const double xx1 = fmax(rect[u][0], rect[t][0]), yy1 = fmax(rect[u][1], rect[t][1]);
                const double xx2 = fmin(rect[u][2], rect[t][2]), yy2 = fmin(rect[u][3], rect[t][3]);
                const double w = fmax(xx2 - xx1, 0.0), h = fmax(yy2 - yy1, 0.0);
                const double inter = w * h;
                const double au = (rect[u][2] - rect[u][0]) * (rect[u][3] - rect[u][1]);
                const double iou = inter / ((au - inter) + ai);
                if (!(iou <= nms_thres)) dead[u] = true;        // IoU.le(overlap) keeps; NaN does not
            }
        }
        count_out[b] = kept;
    }
}

inline bool pr_args_ok(int32_t B, int32_t N, int32_t top_k) {
    return B > 0 && N > 0 && top_k > 0 && top_k <= PR_MAX_TOPK && (int64_t)B * N < (1ll << 31) / 8;
}

}  // namespace

extern "C" size_t vn_rpn_predict_workspace_bytes(int32_t B, int32_t n_anchors) {
    if (!pr_args_ok(B, n_anchors, 1)) return 0;
    return vn_align((size_t)B * sizeof(int32_t)) + vn_align((size_t)B * n_anchors * sizeof(float)) +
           vn_align((size_t)B * n_anchors * sizeof(int32_t));
}

extern "C" int vn_rpn_predict(const float *probs, const float *deltas, const double *anchors, int32_t B, int32_t n_anchors,
                              float score_thres, double nms_thres, int32_t top_k, double anchor_h, float *boxes,
                              float *scores, int32_t *counts, void *workspace, size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(probs && deltas && anchors && boxes && scores && counts && workspace && pr_args_ok(B, n_anchors, top_k));
    if (workspace_bytes < vn_rpn_predict_workspace_bytes(B, n_anchors)) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    char *ws = static_cast<char *>(workspace);
    int32_t *cnt = reinterpret_cast<int32_t *>(ws);
    float *cand_score = reinterpret_cast<float *>(ws + vn_align((size_t)B * sizeof(int32_t)));
    int32_t *cand_idx = reinterpret_cast<int32_t *>(ws + vn_align((size_t)B * sizeof(int32_t)) +
                                                    vn_align((size_t)B * n_anchors * sizeof(float)));
    VN_HIP(hipMemsetAsync(cnt, 0, (size_t)B * sizeof(int32_t), st));
    const dim3 grid((unsigned)((n_anchors + PR_THREADS - 1) / PR_THREADS), (unsigned)B);
    k_pr_filter<<<grid, PR_THREADS, 0, st>>>(probs, n_anchors, score_thres, cnt, cand_score, cand_idx);
    VN_LAUNCH_STATUS();
    k_pr_select_nms<<<B, PR_THREADS, 0, st>>>(deltas, anchors, n_anchors, cnt, cand_score, cand_idx, top_k, nms_thres, anchor_h,
                                             boxes, scores, counts);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
