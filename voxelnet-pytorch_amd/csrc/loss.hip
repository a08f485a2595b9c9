// RPN loss of RPN3D.forward (model.py:309-352) and smooth_L1_loss (loss.py:3-13, its option1*option2 quirk
// included), forward and backward, each as ONE pass over the (B, h, w) anchor sites instead of ~50 elementwise /
// reduction launches.  One thread per site: 2 anchors x (1 score + 7 regression channels).
//   P_b = max(1, sum pos[b]),  N_b = max(1, sum neg[b])                                    (model.py:313-322)
//   cls_pos = -pos * log(p + 1e-6) / P_b ;  cls_neg = -neg * log(1 - p + 1e-6) / N_b       (model.py:340-341)
//   reg     = smooth_L1(delta*posr, tgt*posr) / P_b                                         (model.py:347-349)
//   out[5]  = [alpha*S_pos + beta*S_neg + S_reg, alpha*S_pos + beta*S_neg, S_reg, S_pos, S_neg]
// Layouts: prob (B,2,h,w), delta (B,14,h,w) fp32 NCHW (the module's outputs); pos/neg (B,h,w,2), targets (B,h,w,14)
// fp32 channels-last (utils.generate_targets' arrays, model.py:309).  HBM-bound: ~0.1 KB per site, 70,400 sites.
// Sums: per-thread fp32 -> per-workgroup slab rows -> one workgroup reduces the slab in double (deterministic).
#include "common.h"

namespace {

constexpr int LOSS_THREADS = 256;

constexpr int NORM_CHUNKS = 64;   // workgroups per sample in the normaliser reduction

// part[(b*NORM_CHUNKS + chunk)*2 + {0,1}] = partial sums of pos / neg ; combined in fixed order by k_loss_norm_final
__global__ void __launch_bounds__(LOSS_THREADS) k_loss_norm(const float *__restrict__ pos, const float *__restrict__ neg,
                                                            int64_t per_b, float *__restrict__ part) {
    VN_PRIO_MAIN();
    const int b = blockIdx.x / NORM_CHUNKS, chunk = blockIdx.x % NORM_CHUNKS;
    float sp = 0.f, sn = 0.f;
    for (int64_t i = (int64_t)chunk * LOSS_THREADS + threadIdx.x; i < per_b; i += (int64_t)NORM_CHUNKS * LOSS_THREADS) {
        sp += pos[(int64_t)b * per_b + i];
        sn += neg[(int64_t)b * per_b + i];
    }
    __shared__ float red[2][LOSS_THREADS / 64];
    sp = vn_wave_sum(sp);
    sn = vn_wave_sum(sn);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sp; red[1][threadIdx.x >> 6] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) { a += red[0][w]; c += red[1][w]; }
        part[(int64_t)blockIdx.x * 2] = a;
        part[(int64_t)blockIdx.x * 2 + 1] = c;
    }
}

__global__ void __launch_bounds__(64) k_loss_norm_final(const float *__restrict__ part, float *__restrict__ norm, int B) {
    VN_PRIO_MAIN();
    const int b = blockIdx.x, lane = threadIdx.x;   // NORM_CHUNKS == 64 == one wave
    const float a = vn_wave_sum(part[((int64_t)b * NORM_CHUNKS + lane) * 2]);
    const float c = vn_wave_sum(part[((int64_t)b * NORM_CHUNKS + lane) * 2 + 1]);
    if (lane == 0) {
        norm[b] = fmaxf(a, 1.f);
        norm[B + b] = fmaxf(c, 1.f);
    }
}

struct LossGeom {
    int32_t B, H, W;
    float alpha, beta, sigma2;
};

__device__ __forceinline__ float smooth_l1(float diff, float sigma2, float *ddiff) {
    const float ad = fabsf(diff);
    const float sign = ad < 1.0f / sigma2 ? 1.f : 0.f;
    const float o1 = diff * diff * 0.5f * sigma2;
    const float o2 = ad - 0.5f / sigma2;
    const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);     // d|x|/dx with 0 at 0, as autograd has it
    *ddiff = sigma2 * diff * o2 + o1 * sg + sg * (1.f - sign);
    return o1 * o2 + o2 * (1.f - sign);
}

// upstream gradients of the five outputs as separate device scalars (what autograd hands a five-output Function: no
// stack / zeros kernels to assemble a vector); a NULL pointer stands for 0
struct LossGrads {
    const float *g[5];
};

// MODE 0: the forward sums; 1: the gradients; 2: both in ONE pass over the sites (vn_rpn_loss_fwd_bwd: the same arithmetic in
// the same order per thread, so sums and gradients are bit-identical to the two separate passes)
// HeadRows (vn_rpn_loss_fwd_bwd_rows): a thread holds the 2 + 14 gradients of its site — exactly one row of the (B*S, 16)
// gradient the heads' backward reads (vn_heads_bwd: d_logit = d_prob * p * (1 - p), then the 14 regression gradients) — so
// it writes that row as well and the heads_bwd launch leaves the chain between the loss and the heads' data gradient.
struct HeadRows {
    void *rows;        // NULL: none
    int64_t stride;    // elements per row
    int f32, split;    // fp32 rows | bf16 rows (split: the lo parts at [16 + c], vn_heads_bwd's form)
};

template <int MODE>
__global__ void __launch_bounds__(LOSS_THREADS) k_loss(const float *__restrict__ prob, const float *__restrict__ reg,
                                                       const float *__restrict__ pos, const float *__restrict__ neg,
                                                       const float *__restrict__ tgt, const float *__restrict__ norm,
                                                       LossGeom g, float *__restrict__ slab /* fwd: [blocks][3] */,
                                                       LossGrads gout /* bwd: five device scalars, NULL = 0 */,
                                                       float *__restrict__ d_prob, float *__restrict__ d_reg, HeadRows hr) {
    VN_PRIO_MAIN();
    constexpr bool BWD = MODE != 0, FWD = MODE != 1;
    const int64_t hw = (int64_t)g.H * g.W, sites = hw * g.B;
    const int64_t site = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x;
    float s_pos = 0.f, s_neg = 0.f, s_reg = 0.f;
    if (site < sites) {
        const int b = (int)(site / hw);
        const int64_t yx = site - (int64_t)b * hw;
        const float inv_p = 1.f / norm[b], inv_n = 1.f / norm[g.B + b];
        const float2 ps = *reinterpret_cast<const float2 *>(pos + site * 2);
        const float2 ng = *reinterpret_cast<const float2 *>(neg + site * 2);
        const float pa[2] = {ps.x, ps.y}, na[2] = {ng.x, ng.y};
        float kp = 0.f, kn = 0.f, kr = 0.f;
        float row[16];
        if (BWD) {
            const float gl = gout.g[0] ? *gout.g[0] : 0.f, gc = gout.g[1] ? *gout.g[1] : 0.f, gr = gout.g[2] ? *gout.g[2] : 0.f,
                        gp = gout.g[3] ? *gout.g[3] : 0.f, gn = gout.g[4] ? *gout.g[4] : 0.f;
            kp = g.alpha * (gl + gc) + gp;
            kn = g.beta * (gl + gc) + gn;
            kr = gl + gr;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int64_t pi = ((int64_t)b * 2 + a) * hw + yx;
            const float p = prob[pi];
            if (BWD) {
                const float dp = -kp * pa[a] * inv_p / (p + 1e-6f) + kn * na[a] * inv_n / (1.f - p + 1e-6f);
                d_prob[pi] = dp;
                row[a] = dp * p * (1.0f - p);      // (vn_heads_bwd's expression on the stored values)
            }
            if (FWD) {
                s_pos += -pa[a] * logf(p + 1e-6f) * inv_p;
                s_neg += -na[a] * logf(1.f - p + 1e-6f) * inv_n;
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int c = a * 7 + j;
                const int64_t ri = ((int64_t)b * 14 + c) * hw + yx;
                const float diff = reg[ri] * pa[a] - tgt[site * 14 + c] * pa[a];
                float dd;
                const float l = smooth_l1(diff, g.sigma2, &dd);
                if (BWD) {
                    const float dr = kr * dd * pa[a] * inv_p;
                    d_reg[ri] = dr;
                    row[2 + c] = dr;
                }
                if (FWD) s_reg += l * inv_p;
            }
        }
        if (BWD && hr.rows) {
            if (hr.f32) {
                float *d = static_cast<float *>(hr.rows) + site * hr.stride;
#pragma unroll
                for (int c = 0; c < 16; ++c) d[c] = row[c];
            } else {
                bf16_t *d = static_cast<bf16_t *>(hr.rows) + site * hr.stride;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    bf16_t h, l;
                    vn_split_bf16(row[c], h, l);
                    d[c] = h;
                    if (hr.split) d[16 + c] = l;
                }
            }
        }
    }
    if (FWD) {
        __shared__ float red[3][LOSS_THREADS / 64];
        s_pos = vn_wave_sum(s_pos); s_neg = vn_wave_sum(s_neg); s_reg = vn_wave_sum(s_reg);
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = s_pos; red[1][threadIdx.x >> 6] = s_neg; red[2][threadIdx.x >> 6] = s_reg;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float v = 0.f;
            for (int w = 0; w < LOSS_THREADS / 64; ++w) v += red[threadIdx.x][w];
            slab[(int64_t)blockIdx.x * 3 + threadIdx.x] = v;
        }
    }
}

__global__ void __launch_bounds__(LOSS_THREADS) k_loss_finalize(const float *__restrict__ slab, int nblocks, float alpha,
                                                                float beta, float *__restrict__ out) {
    VN_PRIO_MAIN();
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblocks; i += LOSS_THREADS)
        for (int k = 0; k < 3; ++k) s[k] += (double)slab[(int64_t)i * 3 + k];
    __shared__ double red[3][LOSS_THREADS / 64];
    for (int k = 0; k < 3; ++k) {
        double v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3];
        for (int k = 0; k < 3; ++k) { t[k] = 0.0; for (int w = 0; w < LOSS_THREADS / 64; ++w) t[k] += red[k][w]; }
        const double cls = (double)alpha * t[0] + (double)beta * t[1];
        out[0] = (float)(cls + t[2]);
        out[1] = (float)cls;
        out[2] = (float)t[2];
        out[3] = (float)t[0];
        out[4] = (float)t[1];
    }
}

bool loss_args_ok(int32_t B, int32_t H, int32_t W) { return B > 0 && H > 0 && W > 0 && (int64_t)B * H * W < (1ll << 31); }

}  // namespace

extern "C" size_t vn_rpn_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
    if (!loss_args_ok(B, H, W)) return 0;
    const int64_t blocks = vn_ceil_div((int64_t)B * H * W, LOSS_THREADS);
    size_t slab = sizeof(float) * 3 * (size_t)blocks, part = sizeof(float) * 2 * NORM_CHUNKS * (size_t)B;
    return vn_align(sizeof(float) * 2 * (size_t)B) + vn_align(slab > part ? slab : part);
}

extern "C" int vn_rpn_loss_fwd(const float *prob, const float *delta, const float *pos, const float *neg,
                               const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                               void *workspace, size_t workspace_bytes, float *out5, vnStream stream) {
    VN_CHECK_ARG(prob && delta && pos && neg && targets && workspace && out5 && loss_args_ok(B, H, W) && sigma > 0.f);
    if (workspace_bytes < vn_rpn_loss_workspace_bytes(B, H, W)) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    float *norm = static_cast<float *>(workspace);
    float *slab = reinterpret_cast<float *>(static_cast<char *>(workspace) + vn_align(sizeof(float) * 2 * (size_t)B));
    const int blocks = (int)vn_ceil_div((int64_t)B * H * W, LOSS_THREADS);
    k_loss_norm<<<B * NORM_CHUNKS, LOSS_THREADS, 0, st>>>(pos, neg, (int64_t)H * W * 2, slab);   // slab: scratch here
    VN_LAUNCH_STATUS();
    k_loss_norm_final<<<B, 64, 0, st>>>(slab, norm, B);
    VN_LAUNCH_STATUS();
    const LossGeom g{B, H, W, alpha, beta, sigma * sigma};
    k_loss<0><<<blocks, LOSS_THREADS, 0, st>>>(prob, delta, pos, neg, targets, norm, g, slab, LossGrads{}, nullptr, nullptr,
                                               HeadRows{});
    VN_LAUNCH_STATUS();
    k_loss_finalize<<<1, LOSS_THREADS, 0, st>>>(slab, blocks, alpha, beta, out5);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_rpn_loss_bwd(const float *prob, const float *delta, const float *pos, const float *neg,
                               const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                               const void *workspace, const float *g_loss, const float *g_cls, const float *g_reg,
                               const float *g_cls_pos, const float *g_cls_neg, float *d_prob, float *d_delta,
                               vnStream stream) {
    VN_CHECK_ARG(prob && delta && pos && neg && targets && workspace && d_prob && d_delta && loss_args_ok(B, H, W) &&
                 sigma > 0.f);
    const float *norm = static_cast<const float *>(workspace);   // written by vn_rpn_loss_fwd
    const int blocks = (int)vn_ceil_div((int64_t)B * H * W, LOSS_THREADS);
    const LossGeom g{B, H, W, alpha, beta, sigma * sigma};
    k_loss<1><<<blocks, LOSS_THREADS, 0, vn_stream(stream)>>>(prob, delta, pos, neg, targets, norm, g, nullptr,
                                                                  LossGrads{{g_loss, g_cls, g_reg, g_cls_pos, g_cls_neg}},
                                                                  d_prob, d_delta, HeadRows{});
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// ---- the same loss in three pieces, for a caller that schedules them itself (vn_net_step): the normalisers depend on the
// target maps only (any stream, any time before the pass), ONE pass over the sites gives the forward sums AND the
// gradients, and the five output scalars — nobody's input in a train step — are finished wherever there is room.
// Bit-identical to vn_rpn_loss_fwd + vn_rpn_loss_bwd (tests/test_gpu_loss.py).
extern "C" int vn_rpn_loss_norm(const float *pos, const float *neg, int32_t B, int32_t H, int32_t W, void *workspace,
                                size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(pos && neg && workspace && loss_args_ok(B, H, W));
    if (workspace_bytes < vn_rpn_loss_workspace_bytes(B, H, W)) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    float *norm = static_cast<float *>(workspace);
    float *slab = reinterpret_cast<float *>(static_cast<char *>(workspace) + vn_align(sizeof(float) * 2 * (size_t)B));
    k_loss_norm<<<B * NORM_CHUNKS, LOSS_THREADS, 0, st>>>(pos, neg, (int64_t)H * W * 2, slab);   // slab: scratch here
    VN_LAUNCH_STATUS();
    k_loss_norm_final<<<B, 64, 0, st>>>(slab, norm, B);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

static int loss_fwd_bwd(const float *prob, const float *delta, const float *pos, const float *neg, const float *targets,
                        int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma, void *workspace,
                        size_t workspace_bytes, const LossGrads &up, float *d_prob, float *d_delta, const HeadRows &hr,
                        vnStream stream) {
    VN_CHECK_ARG(prob && delta && pos && neg && targets && workspace && d_prob && d_delta && loss_args_ok(B, H, W) &&
                 sigma > 0.f);
    if (workspace_bytes < vn_rpn_loss_workspace_bytes(B, H, W)) return VN_EWORKSPACE;
    const float *norm = static_cast<const float *>(workspace);   // written by vn_rpn_loss_norm
    float *slab = reinterpret_cast<float *>(static_cast<char *>(workspace) + vn_align(sizeof(float) * 2 * (size_t)B));
    const int blocks = (int)vn_ceil_div((int64_t)B * H * W, LOSS_THREADS);
    const LossGeom g{B, H, W, alpha, beta, sigma * sigma};
    k_loss<2><<<blocks, LOSS_THREADS, 0, vn_stream(stream)>>>(prob, delta, pos, neg, targets, norm, g, slab, up, d_prob, d_delta,
                                                               hr);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_rpn_loss_fwd_bwd(const float *prob, const float *delta, const float *pos, const float *neg,
                                   const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                                   void *workspace, size_t workspace_bytes, const float *g_loss, const float *g_cls,
                                   const float *g_reg, const float *g_cls_pos, const float *g_cls_neg, float *d_prob,
                                   float *d_delta, vnStream stream) {
    return loss_fwd_bwd(prob, delta, pos, neg, targets, B, H, W, alpha, beta, sigma, workspace, workspace_bytes,
                        LossGrads{{g_loss, g_cls, g_reg, g_cls_pos, g_cls_neg}}, d_prob, d_delta, HeadRows{}, stream);
}

extern "C" int vn_rpn_loss_fwd_bwd_rows(const float *prob, const float *delta, const float *pos, const float *neg,
                                        const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta,
                                        float sigma, void *workspace, size_t workspace_bytes, const float *g_loss,
                                        float *d_prob, float *d_delta, void *d_rows, vnDtype d_dtype, int64_t d_stride,
                                        int32_t split, vnStream stream) {
    VN_CHECK_ARG(d_rows && d_stride >= (split ? 32 : 16));
    VN_CHECK_ARG(d_dtype == VN_BF16 || (d_dtype == VN_F32 && !split));
    return loss_fwd_bwd(prob, delta, pos, neg, targets, B, H, W, alpha, beta, sigma, workspace, workspace_bytes,
                        LossGrads{{g_loss, nullptr, nullptr, nullptr, nullptr}}, d_prob, d_delta,
                        HeadRows{d_rows, d_stride, d_dtype == VN_F32, split}, stream);
}

extern "C" int vn_rpn_loss_finalize(const void *workspace, size_t workspace_bytes, int32_t B, int32_t H, int32_t W, float alpha,
                                    float beta, float *out5, vnStream stream) {
    VN_CHECK_ARG(workspace && out5 && loss_args_ok(B, H, W));
    if (workspace_bytes < vn_rpn_loss_workspace_bytes(B, H, W)) return VN_EWORKSPACE;
    const float *slab = reinterpret_cast<const float *>(static_cast<const char *>(workspace) + vn_align(sizeof(float) * 2 * (size_t)B));
    const int blocks = (int)vn_ceil_div((int64_t)B * H * W, LOSS_THREADS);
    k_loss_finalize<<<1, LOSS_THREADS, 0, vn_stream(stream)>>>(slab, blocks, alpha, beta, out5);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
