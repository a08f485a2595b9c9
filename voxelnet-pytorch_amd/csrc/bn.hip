// Train-/eval-mode BatchNorm (+ReLU) over channels-last rows — the batch_norm /
// relu calls at model.py:72,76 (BatchNorm1d), 142,153,165-166 (ConvMD) and 193,198
// (DeConv2d), forward and backward.  HBM-bound streaming kernels: 8 channels per
// lane (16-B bf16 / 2x16-B fp32 accesses), rows distributed over the workgroup,
// per-lane fp32 partials over a bounded number of rows, then double precision for
// the cross-lane / cross-workgroup part (the variance is a difference of large
// numbers when |mean| >> std, which is the case for the first Conv3d: 99 % of its
// output sites equal the bias).
#include "common.h"

namespace {

__device__ __forceinline__ void load8(const void *base, int dtype, int64_t off, float v[8]) {
    if (dtype == VN_F32X3S) {      // split fp32 storage: 32 B per 8 channels = eight hi bf16 parts, then eight lo parts
        const bf16_t *g = reinterpret_cast<const bf16_t *>(static_cast<const float *>(base) + off);
        const bf16x8_t hi = *reinterpret_cast<const bf16x8_t *>(g), lo = *reinterpret_cast<const bf16x8_t *>(g + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)hi[j] + (float)lo[j];
    } else if (dtype == VN_F32) {
        const float4 a = *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + off);
        const float4 b = *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const bf16x8_t t = *reinterpret_cast<const bf16x8_t *>(static_cast<const bf16_t *>(base) + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
    }
}

// BEV fold (model.py:262): the (B,2,H,W,64) conv output row m = ((b*2+d)*HW + r) lives at row (b*HW + r), channels
// d*64.. of the (B,1,H,W,128) tensor the 2-D layers see.  fold = HW (0: no fold), wide = that tensor's row stride.
__device__ __forceinline__ int64_t fold_off(int64_t m, int64_t fold, int64_t wide, int C) {
    const int64_t seg = m / fold, r = m - seg * fold;
    return ((seg >> 1) * fold + r) * wide + (seg & 1) * C;
}

__device__ __forceinline__ void store8(void *base, int dtype, int64_t lo_off, int64_t off, const float v[8]) {
    if (dtype == VN_F32X3S) {      // split fp32 storage (the fp32x3 kernels' operand format: no split work left for them)
        bf16x8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { bf16_t h, l; vn_split_bf16(v[j], h, l); hi[j] = h; lo[j] = l; }
        bf16_t *g = reinterpret_cast<bf16_t *>(static_cast<float *>(base) + off);
        *reinterpret_cast<bf16x8_t *>(g) = hi;
        *reinterpret_cast<bf16x8_t *>(g + 8) = lo;
    } else if (dtype == VN_F32) {
        float *d = static_cast<float *>(base) + off;
        *reinterpret_cast<float4 *>(d) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        bf16x8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { bf16_t h, l; vn_split_bf16(v[j], h, l); hi[j] = h; lo[j] = l; }
        bf16_t *d = static_cast<bf16_t *>(base) + off;
        *reinterpret_cast<bf16x8_t *>(d) = hi;
        if (lo_off) *reinterpret_cast<bf16x8_t *>(d + lo_off) = lo;
    }
}

// block-level column reduction of 2 x 8 floats per thread -> double atomics
__device__ __forceinline__ void block_reduce_atomic(const float s1[8], const float s2[8], int groups, int rpb, int C,
                                                    double *sums) {
    __shared__ float red[256 * 16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x * 16 + j] = s1[j];
        red[threadIdx.x * 16 + 8 + j] = s2[j];
    }
    __syncthreads();
    // thread t < groups*16 sums one (group, slot) column over the rpb row-slices
    for (int t = threadIdx.x; t < groups * 16; t += 256) {
        const int g = t >> 4, slot = t & 15;
        double acc = 0.0;
        for (int r = 0; r < rpb; ++r) acc += (double)red[(r * groups + g) * 16 + slot];
        const int c = g * 8 + (slot & 7);
        atomicAdd(sums + (slot >= 8 ? C : 0) + c, acc);
    }
}

// same reduction, written as one slab row per workgroup: slab[block][2][C] floats (plain stores)
__device__ __forceinline__ void block_reduce_slab(const float s1[8], const float s2[8], int groups, int rpb, int C,
                                                  float *slab_row) {
    __shared__ float red[256 * 16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x * 16 + j] = s1[j];
        red[threadIdx.x * 16 + 8 + j] = s2[j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < groups * 16; t += 256) {
        const int g = t >> 4, slot = t & 15;
        float acc = 0.f;
        for (int r = 0; r < rpb; ++r) acc += red[(r * groups + g) * 16 + slot];
        slab_row[(slot >= 8 ? C : 0) + g * 8 + (slot & 7)] = acc;
    }
}

__global__ void __launch_bounds__(256) k_bn_stats(const void *__restrict__ y, int dtype, int64_t M, int C,
                                                  int64_t stride, int fold, const float *__restrict__ shift,
                                                  double *__restrict__ sums) {
    VN_PRIO_MAIN();
    const int groups = C >> 3, rpb = 256 / groups;
    const int g = threadIdx.x % groups, rr = threadIdx.x / groups;
    const int creal = C / fold;
    float sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = shift ? shift[(g * 8 + j) % creal] : 0.f;
    float s1[8] = {0}, s2[8] = {0};
    if (rr < rpb) {
        for (int64_t m = (int64_t)blockIdx.x * rpb + rr; m < M; m += (int64_t)gridDim.x * rpb) {
            float v[8];
            load8(y, dtype, m * stride + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[j] - sh[j];
                s1[j] += d;
                s2[j] += d * d;
            }
        }
    }
    block_reduce_atomic(s1, s2, groups, rpb, C, sums);
}

__global__ void __launch_bounds__(256) k_bn_finalize(const double *__restrict__ sums, int64_t M, int C, int fold,
                                                     const float *__restrict__ shift, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float *running_mean,
                                                     float *running_var, int training, float momentum, float eps,
                                                     float *__restrict__ stats) {
    VN_PRIO_MAIN();
    const int creal = C / fold;
    for (int c = threadIdx.x; c < creal; c += blockDim.x) {
        double mean, var;
        if (training) {
            double s1 = 0.0, s2 = 0.0;
            for (int f = 0; f < fold; ++f) { s1 += sums[c + f * creal]; s2 += sums[C + c + f * creal]; }
            const double n = (double)M * fold;
            const double ms = s1 / n;
            var = s2 / n - ms * ms;
            if (var < 0.0) var = 0.0;
            mean = ms + (shift ? (double)shift[c] : 0.0);
            if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            if (running_var) {
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float S = gamma[c] * invstd;
        for (int f = 0; f < fold; ++f) {
            const int ce = c + f * creal;
            stats[ce] = (float)mean;
            stats[C + ce] = invstd;
            stats[2 * C + ce] = S;
            stats[3 * C + ce] = beta[c];
        }
    }
}

// one workgroup per channel: sums the slab column pair in double, then the same epilogue as k_bn_finalize
__global__ void __launch_bounds__(256) k_bn_finalize_slab(const float *__restrict__ slab, int64_t rows, int64_t M, int C,
                                                          const float *__restrict__ shift,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *running_mean,
                                                          float *running_var, float momentum, float eps,
                                                          float *__restrict__ stats) {
    VN_PRIO_MAIN();
    // This launch sits on the dependency chain 46 times per step and is pure latency: the per-channel parameters are
    // requested before the slab loop (not after the reduction: one memory round trip less), and the 256 partial sums
    // meet through wave shuffles + one barrier (fixed order: deterministic) instead of an 8-barrier LDS tree.
    __shared__ double r1[4], r2[4];
    const int c = blockIdx.x;
    float p_shift = 0.f, p_gamma = 0.f, p_beta = 0.f, p_rm = 0.f, p_rv = 0.f;
    if (threadIdx.x == 0) {
        p_shift = shift ? shift[c] : 0.f;
        p_gamma = gamma[c];
        p_beta = beta[c];
        if (running_mean) p_rm = running_mean[c];
        if (running_var) p_rv = running_var[c];
    }
    double s1 = 0.0, s2 = 0.0;
    for (int64_t r = threadIdx.x; r < rows; r += 256) {
        s1 += (double)slab[(r * 2 + 0) * C + c];
        s2 += (double)slab[(r * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t1 = (r1[0] + r1[1]) + (r1[2] + r1[3]), t2 = (r2[0] + r2[1]) + (r2[2] + r2[3]);
        const double n = (double)M;
        const double ms = t1 / n;
        double var = t2 / n - ms * ms;
        if (var < 0.0) var = 0.0;
        const double mean = ms + (double)p_shift;
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * p_rm + momentum * mean);
        if (running_var) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            running_var[c] = (float)((1.0 - momentum) * p_rv + momentum * unb);
        }
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        stats[c] = (float)mean;
        stats[C + c] = invstd;
        stats[2 * C + c] = p_gamma * invstd;
        stats[3 * C + c] = p_beta;
    }
}

// value a y tensor of dtype ydt holds where the conv wrote the plain fp32 value x (the first Conv3d's bias fill)
__device__ __forceinline__ float as_stored(float x, int ydt) { return ydt == VN_F32 ? x : (float)(bf16_t)x; }

// FLAGGED (first middle layer only): rows with flag 0 hold inactive[c] in every channel (the conv bias: no occupied voxel
// in their receptive field) — their y is not read.  A template parameter, not a run-time branch: the extra arguments
// and the per-row test cost the plain instantiation 60 % of its speed when they were folded into one kernel (measured).
template <bool FLAGGED>
__global__ void __launch_bounds__(256) k_bn_apply(const void *__restrict__ y, int ydt, int64_t ystride, int64_t M, int C,
                                                  const float *__restrict__ stats, int relu, void *__restrict__ a,
                                                  int adt, int64_t astride, int64_t lo_off, int64_t fold,
                                                  const uint8_t *__restrict__ flags, const float *__restrict__ inactive,
                                                  int hoist) {
    VN_PRIO_MAIN();
    const int groups = C >> 3;
    const int64_t total = M * groups;
    {
        if (256 % groups == 0 && !fold && (FLAGGED || hoist)) {
            // a thread keeps one group of 8 channels: the activation of an inactive row is computed ONCE (the rows without a
            // flag — 90 % of the first layer's — are then 16-B stores of it: the pass is a write stream)
            const int rpb = 256 / groups;
            const int c = (threadIdx.x % groups) << 3, rr = threadIdx.x / groups;
            float mean[8], S[8], be[8], cv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mean[j] = stats[c + j]; S[j] = stats[2 * C + c + j]; be[j] = stats[3 * C + c + j];
                cv[j] = 0.f;
                if constexpr (FLAGGED) {
                    const float z = fmaf(S[j], as_stored(inactive[c + j], ydt) - mean[j], be[j]);
                    cv[j] = (relu & 1) ? fmaxf(z, 0.f) : z;
                    if (relu & 2) cv[j] = (float)(bf16_t)cv[j];
                }
            }
            for (int64_t m = (int64_t)blockIdx.x * rpb + rr; m < M; m += (int64_t)gridDim.x * rpb) {
                float v[8];
                bool inact = false;
                if constexpr (FLAGGED) inact = !flags[m];
                if (inact) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = cv[j];
                } else {
                    load8(y, ydt, m * ystride + c, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float z = fmaf(S[j], v[j] - mean[j], be[j]);
                        v[j] = (relu & 1) ? fmaxf(z, 0.f) : z;
                        if (relu & 2) v[j] = (float)(bf16_t)v[j];
                    }
                }
                store8(a, adt, lo_off, m * astride + c, v);
            }
            return;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / groups;
        const int c = (int)(i - m * groups) << 3;
        float v[8];
        bool inact = false;
        if constexpr (FLAGGED) inact = !flags[m];
        if (inact) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = as_stored(inactive[c + j], ydt);
        } else {
            load8(y, ydt, m * ystride + c, v);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = fmaf(stats[2 * C + c + j], v[j] - stats[c + j], stats[3 * C + c + j]);
            v[j] = (relu & 1) ? fmaxf(z, 0.f) : z;
            if (relu & 2) v[j] = (float)(bf16_t)v[j];
        }
        store8(a, adt, lo_off, (fold ? fold_off(m, fold, astride, C) : m * astride) + c, v);
    }
}

template <bool FLAGGED>
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(const void *__restrict__ da, int dadt, int64_t dastride,
                                                       const void *__restrict__ y, int ydt, int64_t ystride, int64_t M,
                                                       int C, const float *__restrict__ stats, int relu,
                                                       double *__restrict__ sums, float *__restrict__ slab, int64_t fold,
                                                       const uint8_t *__restrict__ flags, const float *__restrict__ inactive) {
    VN_PRIO_MAIN();
    const int groups = C >> 3, rpb = 256 / groups;
    const int g = threadIdx.x % groups, rr = threadIdx.x / groups;
    float mean[8], invstd[8], S[8], be[8], yin[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = g * 8 + j;
        mean[j] = stats[c]; invstd[j] = stats[C + c]; S[j] = stats[2 * C + c]; be[j] = stats[3 * C + c];
        yin[j] = 0.f;
        if constexpr (FLAGGED) yin[j] = as_stored(inactive[c], ydt);      // what y holds in the rows with flag 0 (not read there)
    }
    float s1[8] = {0}, s2[8] = {0};
    if (rr < rpb) {
        // two rows in flight per lane (the loads of both are issued before either is used)
        const int64_t step = (int64_t)gridDim.x * rpb;
        int64_t m = (int64_t)blockIdx.x * rpb + rr;
        for (; m + step < M; m += 2 * step) {
            float yv[2][8], dv[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t mu = m + u * step;
                bool inact = false;
                if constexpr (FLAGGED) inact = !flags[mu];
                if (inact) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) yv[u][j] = yin[j];
                } else {
                    load8(y, ydt, mu * ystride + g * 8, yv[u]);
                }
                load8(da, dadt, (fold ? fold_off(mu, fold, dastride, C) : mu * dastride) + g * 8, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d0 = yv[u][j] - mean[j];
                    const float z = fmaf(S[j], d0, be[j]);
                    const float dz = (!relu || z > 0.f) ? dv[u][j] : 0.f;
                    s1[j] += dz;
                    s2[j] += dz * (d0 * invstd[j]);
                }
        }
        for (; m < M; m += step) {
            float yv[8], dv[8];
            bool inact = false;
            if constexpr (FLAGGED) inact = !flags[m];
            if (inact) {
#pragma unroll
                for (int j = 0; j < 8; ++j) yv[j] = yin[j];
            } else {
                load8(y, ydt, m * ystride + g * 8, yv);
            }
            load8(da, dadt, (fold ? fold_off(m, fold, dastride, C) : m * dastride) + g * 8, dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d0 = yv[j] - mean[j];
                const float z = fmaf(S[j], d0, be[j]);
                const float dz = (!relu || z > 0.f) ? dv[j] : 0.f;
                s1[j] += dz;
                s2[j] += dz * (d0 * invstd[j]);
            }
        }
    }
    if (slab) block_reduce_slab(s1, s2, groups, rpb, C, slab + (size_t)blockIdx.x * 2 * C);
    else block_reduce_atomic(s1, s2, groups, rpb, C, sums);
}

// one workgroup per channel: coef from the slab of k_bn_bwd_reduce
__global__ void __launch_bounds__(256) k_bn_bwd_finalize_slab(const float *__restrict__ slab, int rows, int64_t M, int C,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ stats, float *__restrict__ coef,
                                                              float *__restrict__ d_gamma, float *__restrict__ d_beta) {
    VN_PRIO_MAIN();
    __shared__ double r1[4], r2[4];   // (same shape as k_bn_finalize_slab: parameters first, shuffles + one barrier)
    const int c = blockIdx.x;
    float invstd = 0.f, p_gamma = 0.f;
    if (threadIdx.x == 0) { invstd = stats[C + c]; p_gamma = gamma[c]; }
    double s1 = 0.0, s2 = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        s1 += (double)slab[((size_t)r * 2 + 0) * C + c];
        s2 += (double)slab[((size_t)r * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t1 = (r1[0] + r1[1]) + (r1[2] + r1[3]), t2 = (r2[0] + r2[1]) + (r2[2] + r2[3]);
        const double n = (double)M;
        const float S = p_gamma * invstd;
        if (d_gamma) d_gamma[c] = (float)t2;
        if (d_beta) d_beta[c] = (float)t1;
        coef[c] = S;
        coef[C + c] = -S * invstd * (float)(t2 / n);
        coef[2 * C + c] = -S * (float)(t1 / n);
    }
}

__global__ void __launch_bounds__(256) k_bn_bwd_finalize(const double *__restrict__ sums, int64_t M, int C, int fold,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ stats, float *__restrict__ coef,
                                                         float *__restrict__ d_gamma, float *__restrict__ d_beta) {
    VN_PRIO_MAIN();
    const int creal = C / fold;
    for (int c = threadIdx.x; c < creal; c += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
        for (int f = 0; f < fold; ++f) { s1 += sums[c + f * creal]; s2 += sums[C + c + f * creal]; }
        const double n = (double)M * fold;
        const float invstd = stats[C + c];
        const float S = gamma[c] * invstd;
        const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
        if (d_gamma) d_gamma[c] = (float)s2;
        if (d_beta) d_beta[c] = (float)s1;
        for (int f = 0; f < fold; ++f) {
            const int ce = c + f * creal;
            coef[ce] = S;
            coef[C + ce] = -S * invstd * m2;
            coef[2 * C + ce] = -S * m1;
        }
    }
}

// A thread keeps ONE group of 8 channels (256 % groups == 0 for the widths of this network, else the per-iteration
// path), so the six per-channel constants live in registers and a row costs two 16-B loads and one store.
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const void *__restrict__ da, int dadt, int64_t dastride,
                                                      const void *__restrict__ y, int ydt, int64_t ystride, int64_t M,
                                                      int C, const float *__restrict__ stats,
                                                      const float *__restrict__ coef, int relu, void *__restrict__ dy,
                                                      int dydt, int64_t dystride, int64_t lo_off,
                                                      const uint8_t *__restrict__ flags, int64_t fold) {
    VN_PRIO_MAIN();
    // flags != NULL: rows with flag 0 are skipped (their dy is never read by the caller's consumers)
    const int groups = C >> 3;
    if (256 % groups == 0) {
        const int rpb = 256 / groups;
        const int c = (threadIdx.x % groups) << 3, rr = threadIdx.x / groups;
        float mean[8], S[8], be[8], c0[8], c1[8], c2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mean[j] = stats[c + j]; S[j] = stats[2 * C + c + j]; be[j] = stats[3 * C + c + j];
            c0[j] = coef[c + j]; c1[j] = coef[C + c + j]; c2[j] = coef[2 * C + c + j];
        }
        for (int64_t m = (int64_t)blockIdx.x * rpb + rr; m < M; m += (int64_t)gridDim.x * rpb) {
            if (flags && !flags[m]) continue;
            float yv[8], dv[8], o[8];
            load8(y, ydt, m * ystride + c, yv);
            load8(da, dadt, (fold ? fold_off(m, fold, dastride, C) : m * dastride) + c, dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d0 = yv[j] - mean[j];
                const float z = fmaf(S[j], d0, be[j]);
                const float dz = (!relu || z > 0.f) ? dv[j] : 0.f;
                o[j] = fmaf(c0[j], dz, fmaf(c1[j], d0, c2[j]));
            }
            store8(dy, dydt, lo_off, m * dystride + c, o);
        }
        return;
    }
    const int64_t total = M * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / groups;
        if (flags && !flags[m]) continue;
        const int c = (int)(i - m * groups) << 3;
        float yv[8], dv[8], o[8];
        load8(y, ydt, m * ystride + c, yv);
        load8(da, dadt, (fold ? fold_off(m, fold, dastride, C) : m * dastride) + c, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d0 = yv[j] - stats[c + j];
            const float z = fmaf(stats[2 * C + c + j], d0, stats[3 * C + c + j]);
            const float dz = (!relu || z > 0.f) ? dv[j] : 0.f;
            o[j] = fmaf(coef[c + j], dz, fmaf(coef[C + c + j], d0, coef[2 * C + c + j]));
        }
        store8(dy, dydt, lo_off, m * dystride + c, o);
    }
}

// dy = c0 dz + c1 (y - mean) + c2 at the rows of an explicit list only ((b,d,h,w) int64 coordinates of the dense
// (B,D,H,W,C) tensors, *count valid entries): the first middle layer's gradient kernels read dy at its active sites only
__global__ void __launch_bounds__(256) k_bn_bwd_apply_list(const void *__restrict__ da, int dadt, const void *__restrict__ y,
                                                           int ydt, int C, int D, int H, int W,
                                                           const float *__restrict__ stats, const float *__restrict__ coef,
                                                           int relu, void *__restrict__ dy, int dydt,
                                                           const int64_t *__restrict__ list, const int32_t *__restrict__ count,
                                                           int64_t cap, int da_compact) {
    VN_PRIO_MAIN();
    const int groups = C >> 3, rpb = 256 / groups;
    const int c = (threadIdx.x % groups) << 3, rr = threadIdx.x / groups;
    if (rr >= rpb) return;
    int64_t n = count ? (int64_t)count[0] : cap;
    if (n > cap) n = cap;
    float mean[8], S[8], be[8], c0[8], c1[8], c2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mean[j] = stats[c + j]; S[j] = stats[2 * C + c + j]; be[j] = stats[3 * C + c + j];
        c0[j] = coef[c + j]; c1[j] = coef[C + c + j]; c2[j] = coef[2 * C + c + j];
    }
    for (int64_t e = (int64_t)blockIdx.x * rpb + rr; e < n; e += (int64_t)gridDim.x * rpb) {
        const int64_t *rc = list + e * 4;
        const int64_t m = ((rc[0] * D + rc[1]) * H + rc[2]) * W + rc[3];
        float yv[8], dv[8], o[8];
        load8(y, ydt, m * C + c, yv);
        load8(da, dadt, (da_compact ? e : m) * C + c, dv);     // da_compact: row e of a [cap][C] matrix in list order
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d0 = yv[j] - mean[j];
            const float z = fmaf(S[j], d0, be[j]);
            const float dz = (!relu || z > 0.f) ? dv[j] : 0.f;
            o[j] = fmaf(c0[j], dz, fmaf(c1[j], d0, c2[j]));
        }
        store8(dy, dydt, 0, m * C + c, o);
    }
}

// ---- the first middle layer's BatchNorm backward WITHOUT the dense gradient of its activation ------------------------
// Its output y holds inactive[c] (the conv bias) at the ~90 % of sites no occupied voxel reaches, and its own gradient
// kernels read dy at the active sites only.  So the next layer's data gradient `da` is computed at the listed (active)
// sites only ([cap][C] rows in list order: vn_conv_gather_gemm_rows), and the BatchNorm sums over ALL sites are
//     sum dz       = sum_active dz       + mask_c          (T_c - sum_active da)
//     sum dz xhat  = sum_active dz xhat  + mask_c xhat_c   (T_c - sum_active da)
// (mask_c / xhat_c: ReLU mask and normalised value of the constant inactive[c]) with T_c = the sum of da over ALL
// sites, which is linear in the next layer's dy:  T = sum_taps W_tap^T . (sum of dy over the sites whose tap target
// lies inside the grid) — nine box sums of dy (vn_box_col_sums) instead of a dense 3x3x3 data gradient.

// slab[block][3][C]: sum_active dz, sum_active dz*xhat, sum_active da
__global__ void __launch_bounds__(256) k_bn_bwd_reduce_list(const void *__restrict__ da, int dadt, const void *__restrict__ y,
                                                            int ydt, int C, int D, int H, int W,
                                                            const float *__restrict__ stats, int relu,
                                                            float *__restrict__ slab, const int64_t *__restrict__ list,
                                                            const int32_t *__restrict__ count, int64_t cap) {
    VN_PRIO_MAIN();
    __shared__ float red[256 * 24];
    const int groups = C >> 3, rpb = 256 / groups;
    const int g = threadIdx.x % groups, c = g << 3, rr = threadIdx.x / groups;
    int64_t n = count ? (int64_t)count[0] : cap;
    if (n > cap) n = cap;
    float mean[8], invstd[8], S[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = stats[c + j]; invstd[j] = stats[C + c + j]; S[j] = stats[2 * C + c + j]; be[j] = stats[3 * C + c + j]; }
    float s1[8] = {0}, s2[8] = {0}, s3[8] = {0};
    if (rr < rpb) {
        for (int64_t e = (int64_t)blockIdx.x * rpb + rr; e < n; e += (int64_t)gridDim.x * rpb) {
            const int64_t *rc = list + e * 4;
            const int64_t m = ((rc[0] * D + rc[1]) * H + rc[2]) * W + rc[3];
            float yv[8], dv[8];
            load8(y, ydt, m * C + c, yv);
            load8(da, dadt, e * C + c, dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d0 = yv[j] - mean[j];
                const float z = fmaf(S[j], d0, be[j]);
                const float dz = (!relu || z > 0.f) ? dv[j] : 0.f;
                s1[j] += dz;
                s2[j] += dz * (d0 * invstd[j]);
                s3[j] += dv[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x * 24 + j] = s1[j];
        red[threadIdx.x * 24 + 8 + j] = s2[j];
        red[threadIdx.x * 24 + 16 + j] = s3[j];
    }
    __syncthreads();
    float *row = slab + (size_t)blockIdx.x * 3 * C;
    for (int t = threadIdx.x; t < groups * 24; t += 256) {
        const int gg = t / 24, slot = t - gg * 24;
        float acc = 0.f;
        for (int r = 0; r < rpb; ++r) acc += red[(r * groups + gg) * 24 + slot];
        row[(slot >> 3) * C + gg * 8 + (slot & 7)] = acc;
    }
}

// one workgroup per channel: the three slab columns in double, then the inactive sites' closed form and the coefficients
__global__ void __launch_bounds__(256) k_bn_bwd_finalize_list(const float *__restrict__ slab, int rows, int64_t M, int C,
                                                              const float *__restrict__ gamma, const float *__restrict__ stats,
                                                              const float *__restrict__ total, const float *__restrict__ inactive,
                                                              int ydt, int relu, float *__restrict__ coef,
                                                              float *__restrict__ d_gamma, float *__restrict__ d_beta) {
    VN_PRIO_MAIN();
    __shared__ double r1[256], r2[256], r3[256];
    const int c = blockIdx.x;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        a1 += (double)slab[((size_t)r * 3 + 0) * C + c];
        a2 += (double)slab[((size_t)r * 3 + 1) * C + c];
        a3 += (double)slab[((size_t)r * 3 + 2) * C + c];
    }
    r1[threadIdx.x] = a1; r2[threadIdx.x] = a2; r3[threadIdx.x] = a3;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; r3[threadIdx.x] += r3[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = stats[c], invstd = stats[C + c], Sc = stats[2 * C + c], be = stats[3 * C + c];
        const float d0 = as_stored(inactive[c], ydt) - mean;           // the same fp32 expressions as the row kernels
        const float z = fmaf(Sc, d0, be);
        const double rest = (!relu || z > 0.f) ? (double)total[c] - r3[0] : 0.0;   // sum of dz over the inactive sites
        const double s1 = r1[0] + rest, s2 = r2[0] + rest * (double)(d0 * invstd);
        const double n = (double)M;
        const float S = gamma[c] * invstd;
        if (d_gamma) d_gamma[c] = (float)s2;
        if (d_beta) d_beta[c] = (float)s1;
        coef[c] = S;
        coef[C + c] = -S * invstd * (float)(s2 / n);
        coef[2 * C + c] = -S * (float)(s1 / n);
    }
}

// Box sums of a dense (B,D,H,W,C) rows tensor per channel: blocks [0, nb) sum all rows (grid-stride; slab[b][C]); blocks
// nb + k*BOX_EB .. +BOX_EB sum edge k: 0 h=0, 1 h=H-1, 2 w=0, 3 w=W-1, 4..7 the corners (0,0) (0,W-1) (H-1,0) (H-1,W-1)
// (over all b, d).
constexpr int BOX_EB = 16;
__global__ void __launch_bounds__(256) k_box_partials(const void *__restrict__ x, int dt, int B, int D, int H, int W, int C,
                                                      int nb, float *__restrict__ slab) {
    VN_PRIO_MAIN();
    __shared__ float red[256 * 8];
    const int groups = C >> 3, rpb = 256 / groups;
    const int g = threadIdx.x % groups, c = g << 3, rr = threadIdx.x / groups;
    float s[8] = {0};
    if (rr < rpb) {
        if ((int)blockIdx.x < nb) {
            const int64_t M = (int64_t)B * D * H * W, step = (int64_t)nb * rpb;
            int64_t m = (int64_t)blockIdx.x * rpb + rr;
            for (; m + 3 * step < M; m += 4 * step) {      // four rows in flight
                float v0[8], v1[8], v2[8], v3[8];
                load8(x, dt, m * C + c, v0);
                load8(x, dt, (m + step) * C + c, v1);
                load8(x, dt, (m + 2 * step) * C + c, v2);
                load8(x, dt, (m + 3 * step) * C + c, v3);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
            }
            for (; m < M; m += step) {
                float v[8];
                load8(x, dt, m * C + c, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
            }
        } else {
            const int k = (blockIdx.x - nb) / BOX_EB, part = (blockIdx.x - nb) % BOX_EB;     // BOX_EB blocks share an edge
            const int len = k < 2 ? W : (k < 4 ? H : 1);
            const int64_t total = (int64_t)B * D * len;
            for (int64_t i = (int64_t)part * rpb + rr; i < total; i += (int64_t)BOX_EB * rpb) {
                const int64_t bd = i / len;
                const int t = (int)(i - bd * len);
                int h, w;
                if (k == 0) { h = 0; w = t; } else if (k == 1) { h = H - 1; w = t; }
                else if (k == 2) { h = t; w = 0; } else if (k == 3) { h = t; w = W - 1; }
                else { h = (k & 2) ? H - 1 : 0; w = (k & 1) ? W - 1 : 0; }
                float v[8];
                load8(x, dt, ((bd * H + h) * W + w) * C + c, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = s[j];
    __syncthreads();
    for (int t = threadIdx.x; t < C; t += 256) {
        float acc = 0.f;
        for (int r = 0; r < rpb; ++r) acc += red[(r * groups + (t >> 3)) * 8 + (t & 7)];
        slab[(size_t)blockIdx.x * C + t] = acc;
    }
}

// T[ci] = sum over taps (kd,kh,kw) and co of  Box(kh,kw)[co] * w[co][ci][kd][kh][kw]   (w: the torch Conv3d weight, fp32), with
// Box(kh,kw) = Tot - [kh=0] H0 - [kh=2] H1 - [kw=0] W0 - [kw=2] W1 + the corner both exclusions removed twice: the sum of dy
// over the sites whose tap target (h + kh - 1, w + kw - 1) lies inside the H x W grid (3x3, stride 1, padding 1 in H/W; every
// depth tap in range: no padding in D).  One workgroup per input channel ci (every workgroup re-reduces the small slab).
__global__ void __launch_bounds__(256) k_box_total(const float *__restrict__ slab, int nb, int Co, int Ci, int kD,
                                                   const float *__restrict__ w, int w_bf16, float *__restrict__ total) {
    VN_PRIO_MAIN();
    __shared__ double part[4][256];    // [row quarter][co]
    __shared__ double edge[8][256];
    __shared__ double box[9][256];     // [3*hc + wc][co]: hc / wc 0 = exclude first, 1 = all, 2 = exclude last
    __shared__ double red[256];
    const int ci = blockIdx.x;
    // Tot[co]: nb slab rows, four row quarters in parallel (Co <= 64 per pass of 256 threads; Co <= 256 in all)
    for (int base = 0; base < Co; base += 64) {
        const int co = base + (threadIdx.x & 63), q = threadIdx.x >> 6;
        double a = 0.0;
        if (co < Co) {
            int r = q;
            for (; r + 28 < nb; r += 32) {       // eight loads in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(r + 4 * u) * Co + co];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)v[u];
            }
            for (; r < nb; r += 4) a += (double)slab[(size_t)r * Co + co];
            part[q][co] = a;
        }
    }
    // the eight edge sums: BOX_EB slab rows each, two edges per row quarter, all loads of an edge issued together
    for (int base = 0; base < Co; base += 64) {
        const int co = base + (threadIdx.x & 63), q = threadIdx.x >> 6;
        if (co < Co)
            for (int k = 2 * q; k < 2 * q + 2; ++k) {
                float v[BOX_EB];
#pragma unroll
                for (int j = 0; j < BOX_EB; ++j) v[j] = slab[(size_t)(nb + k * BOX_EB + j) * Co + co];
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < BOX_EB; ++j) a += (double)v[j];
                edge[k][co] = a;
            }
    }
    __syncthreads();
    for (int co = threadIdx.x; co < Co; co += 256) {
        const double tot = (part[0][co] + part[1][co]) + (part[2][co] + part[3][co]);
        double e[8];
        for (int k = 0; k < 8; ++k) e[k] = edge[k][co];
        for (int hc = 0; hc < 3; ++hc)
            for (int wc = 0; wc < 3; ++wc) {
                double v = tot;
                if (hc == 0) v -= e[0];
                if (hc == 2) v -= e[1];
                if (wc == 0) v -= e[2];
                if (wc == 2) v -= e[3];
                if (hc == 0 && wc == 0) v += e[4];
                if (hc == 0 && wc == 2) v += e[5];
                if (hc == 2 && wc == 0) v += e[6];
                if (hc == 2 && wc == 2) v += e[7];
                box[hc * 3 + wc][co] = v;
            }
    }
    __syncthreads();
    double t = 0.0;
    for (int co = threadIdx.x; co < Co; co += 256) {
        const float *wp = w + ((size_t)co * Ci + ci) * kD * 9;
        for (int kd = 0; kd < kD; ++kd)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const float wv = wp[(kd * 3 + kh) * 3 + kw];      // (the data-gradient kernels read bf16-rounded weights)
                    t += box[kh * 3 + kw][co] * (double)(w_bf16 ? (float)(bf16_t)wv : wv);
                }
    }
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[ci] = (float)red[0];
}

inline unsigned gs_blocks(int64_t total, int per_block, int cap) {
    int64_t b = vn_ceil_div(total, per_block);
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}
// Rows per lane of the backward reduction's workgroups: at most 4 (two in flight), and fewer on the small layers so
// that a launch has ~1000 workgroups — those launches are latency-bound (one HBM round trip per row and lane), not
// bandwidth-bound.  The slab has one row per workgroup (vn_bn_bwd_slab_rows uses the same rule).
inline int bwd_rpl(int64_t M, int rpb) {
    const int64_t r = M / ((int64_t)rpb * 1024);
    return r < 1 ? 1 : (r > 4 ? 4 : (int)r);
}
// 8-channel groups per lane of the backward apply's grid (same reasoning)
inline int apply_epl(int64_t total) {
    const int64_t r = total / (256 * 2048);
    return r < 1 ? 1 : (r > 4 ? 4 : (int)r);
}
inline bool rows_ok(int C, int64_t stride) { return C >= 8 && (C & 7) == 0 && C <= 2048 && (stride & 7) == 0; }

}  // namespace

extern "C" int vn_bn_stats(const void *y, vnDtype dtype, int64_t M, int32_t C, int64_t stride, int32_t fold,
                           const float *shift, double *sums, vnStream stream) {
    VN_CHECK_ARG(sums && M >= 0 && fold >= 1 && rows_ok(C, stride) && C % fold == 0);
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(y);
    const int rpb = 256 / (C >> 3);
    k_bn_stats<<<gs_blocks(M, rpb * 8, 2048), 256, 0, vn_stream(stream)>>>(y, (int)dtype, M, C, stride, fold, shift, sums);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_finalize(const double *sums, int64_t M, int32_t C, int32_t fold, const float *shift,
                              const float *gamma, const float *beta, float *running_mean, float *running_var,
                              int32_t training, float momentum, float eps, float *stats, vnStream stream) {
    VN_CHECK_ARG(gamma && beta && stats && C > 0 && fold >= 1 && C % fold == 0);
    VN_CHECK_ARG(training ? (sums != nullptr && M > 0) : (running_mean && running_var));
    k_bn_finalize<<<1, 256, 0, vn_stream(stream)>>>(sums, M, C, fold, shift, gamma, beta, running_mean, running_var,
                                                    training, momentum, eps, stats);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_finalize_slab(const float *slab, int64_t slab_rows, int64_t M, int32_t C, const float *shift,
                                   const float *gamma, const float *beta, float *running_mean, float *running_var,
                                   float momentum, float eps, float *stats, vnStream stream) {
    VN_CHECK_ARG(slab && gamma && beta && stats && slab_rows > 0 && M > 0 && C > 0);
    k_bn_finalize_slab<<<C, 256, 0, vn_stream(stream)>>>(slab, slab_rows, M, C, shift, gamma, beta, running_mean,
                                                         running_var, momentum, eps, stats);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_apply(const void *y, vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                           int32_t relu, void *a, vnDtype a_dtype, int64_t a_stride, int64_t lo_off, vnStream stream) {
    VN_CHECK_ARG(M >= 0 && rows_ok(C, y_stride) && (a_stride & 7) == 0 && (lo_off & 7) == 0 && lo_off >= 0 && (!lo_off || a_dtype == VN_BF16));
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(y && a && stats);
    k_bn_apply<false><<<gs_blocks(M * (C >> 3), 256, 8192), 256, 0, vn_stream(stream)>>>(y, (int)y_dtype, y_stride, M, C, stats,
                                                                                          relu, a, (int)a_dtype, a_stride, lo_off, 0,
                                                                                          nullptr, nullptr, 1);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// vn_bn_apply for a conv output whose rows with row_flags[m] == 0 all hold `inactive` (float[C], e.g. the conv bias of
// the first middle layer at the ~90 % of sites without an occupied voxel in reach): those rows are written without
// reading y.  Same values as vn_bn_apply.
extern "C" int vn_bn_apply_flagged(const void *y, vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                                   int32_t relu, void *a, vnDtype a_dtype, int64_t a_stride, const uint8_t *row_flags,
                                   const float *inactive, vnStream stream) {
    VN_CHECK_ARG(M >= 0 && rows_ok(C, y_stride) && (a_stride & 7) == 0);
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(y && a && stats && row_flags && inactive);
    k_bn_apply<true><<<gs_blocks(M * (C >> 3), 256, 8192), 256, 0, vn_stream(stream)>>>(y, (int)y_dtype, y_stride, M, C, stats,
                                                                                         relu, a, (int)a_dtype, a_stride, 0, 0,
                                                                                         row_flags, inactive, 1);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_bwd_reduce(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y, vnDtype y_dtype,
                                int64_t y_stride, int64_t M, int32_t C, const float *stats, int32_t relu, double *sums,
                                vnStream stream) {
    VN_CHECK_ARG(sums && M >= 0 && rows_ok(C, y_stride) && (da_stride & 7) == 0);
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(da && y && stats);
    const int rpb = 256 / (C >> 3);
    k_bn_bwd_reduce<false><<<gs_blocks(M, rpb * bwd_rpl(M, rpb), 2048), 256, 0, vn_stream(stream)>>>(da, (int)da_dtype, da_stride, y,
                                                                                 (int)y_dtype, y_stride, M, C, stats,
                                                                                 relu, sums, nullptr, 0, nullptr, nullptr);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int64_t vn_bn_bwd_slab_rows(int64_t M, int32_t C) {
    if (M <= 0 || !rows_ok(C, 8)) return 0;
    return gs_blocks(M, (256 / (C >> 3)) * bwd_rpl(M, 256 / (C >> 3)), 2048);
}

extern "C" int vn_bn_bwd_reduce_slab(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y, vnDtype y_dtype,
                                     int64_t y_stride, int64_t M, int32_t C, const float *stats, int32_t relu,
                                     float *slab, vnStream stream) {
    VN_CHECK_ARG(slab && M > 0 && rows_ok(C, y_stride) && (da_stride & 7) == 0 && da && y && stats);
    const int rpb = 256 / (C >> 3);
    k_bn_bwd_reduce<false><<<gs_blocks(M, rpb * bwd_rpl(M, rpb), 2048), 256, 0, vn_stream(stream)>>>(da, (int)da_dtype, da_stride, y,
                                                                                 (int)y_dtype, y_stride, M, C, stats,
                                                                                 relu, nullptr, slab, 0, nullptr, nullptr);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// vn_bn_bwd_reduce_slab for a y whose rows with row_flags[m] == 0 all hold `inactive` (see vn_bn_apply_flagged): y is
// read at the flagged rows only; da is dense.  Same sums.
extern "C" int vn_bn_bwd_reduce_slab_flagged(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y,
                                             vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                                             int32_t relu, float *slab, const uint8_t *row_flags, const float *inactive,
                                             vnStream stream) {
    VN_CHECK_ARG(slab && M > 0 && rows_ok(C, y_stride) && (da_stride & 7) == 0 && da && y && stats && row_flags && inactive);
    const int rpb = 256 / (C >> 3);
    k_bn_bwd_reduce<true><<<gs_blocks(M, rpb * bwd_rpl(M, rpb), 2048), 256, 0, vn_stream(stream)>>>(da, (int)da_dtype, da_stride, y,
                                                                                 (int)y_dtype, y_stride, M, C, stats,
                                                                                 relu, nullptr, slab, 0, row_flags, inactive);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_bwd_finalize_slab(const float *slab, int64_t slab_rows, int64_t M, int32_t C, const float *gamma,
                                       const float *stats, float *coef, float *d_gamma, float *d_beta,
                                       vnStream stream) {
    VN_CHECK_ARG(slab && gamma && stats && coef && slab_rows > 0 && M > 0 && C > 0);
    k_bn_bwd_finalize_slab<<<C, 256, 0, vn_stream(stream)>>>(slab, (int)slab_rows, M, C, gamma, stats, coef, d_gamma,
                                                             d_beta);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_bwd_finalize(const double *sums, int64_t M, int32_t C, int32_t fold, const float *gamma,
                                  const float *stats, float *coef, float *d_gamma, float *d_beta, vnStream stream) {
    VN_CHECK_ARG(sums && gamma && stats && coef && M > 0 && C > 0 && fold >= 1 && C % fold == 0);
    k_bn_bwd_finalize<<<1, 256, 0, vn_stream(stream)>>>(sums, M, C, fold, gamma, stats, coef, d_gamma, d_beta);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_bwd_apply(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y, vnDtype y_dtype,
                               int64_t y_stride, int64_t M, int32_t C, const float *stats, const float *coef,
                               int32_t relu, void *dy, vnDtype dy_dtype, int64_t dy_stride, int64_t lo_off,
                               vnStream stream) {
    VN_CHECK_ARG(M >= 0 && rows_ok(C, y_stride) && (da_stride & 7) == 0 && (dy_stride & 7) == 0);
    VN_CHECK_ARG(lo_off >= 0 && (lo_off & 7) == 0 && (!lo_off || dy_dtype == VN_BF16));
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(da && y && stats && coef && dy);
    k_bn_bwd_apply<<<gs_blocks(M * (C >> 3), 256 * apply_epl(M * (C >> 3)), 8192), 256, 0, vn_stream(stream)>>>(
        da, (int)da_dtype, da_stride, y, (int)y_dtype, y_stride, M, C, stats, coef, relu, dy, (int)dy_dtype, dy_stride,
        lo_off, nullptr, 0);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// Row-flag variant for the first middle layer, whose weight- and data-gradient only gather dy at the sites with an
// occupied voxel in their receptive field (~10 % of 1.4 M rows): rows with flag 0 are skipped.
extern "C" int vn_bn_bwd_apply_flagged(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y, vnDtype y_dtype,
                                       int64_t y_stride, int64_t M, int32_t C, const float *stats, const float *coef,
                                       int32_t relu, void *dy, vnDtype dy_dtype, int64_t dy_stride,
                                       const uint8_t *row_flags, vnStream stream) {
    VN_CHECK_ARG(M >= 0 && rows_ok(C, y_stride) && (da_stride & 7) == 0 && (dy_stride & 7) == 0);
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(da && y && stats && coef && dy && row_flags);
    k_bn_bwd_apply<<<gs_blocks(M * (C >> 3), 256 * apply_epl(M * (C >> 3)), 8192), 256, 0, vn_stream(stream)>>>(
        da, (int)da_dtype, da_stride, y, (int)y_dtype, y_stride, M, C, stats, coef, relu, dy, (int)dy_dtype, dy_stride, 0,
        row_flags, 0);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// ---- BEV-fold variants for the last Conv3d (model.py:262: its (B,2,H,W,64) output is the (B,1,H,W,128) input of
// block1, channel = d*64 + c): ONE launch each instead of one per (batch, depth) slice.  `hw` = H*W; the folded
// tensor (a / da) has row stride wide_stride (= 128); y / dy are the plain (B*2*H*W, 64) rows.
extern "C" int vn_bn_apply_bev(const void *y, vnDtype y_dtype, int64_t M, int32_t C, int64_t hw, const float *stats,
                               int32_t relu, void *a, vnDtype a_dtype, int64_t wide_stride, vnStream stream) {
    VN_CHECK_ARG(M > 0 && rows_ok(C, C) && hw > 0 && M % (2 * hw) == 0 && wide_stride >= 2 * C && (wide_stride & 7) == 0);
    VN_CHECK_ARG(y && a && stats);
    k_bn_apply<false><<<gs_blocks(M * (C >> 3), 256, 8192), 256, 0, vn_stream(stream)>>>(y, (int)y_dtype, C, M, C, stats, relu, a,
                                                                                          (int)a_dtype, wide_stride, 0, hw, nullptr, nullptr, 0);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_bwd_reduce_slab_bev(const void *da, vnDtype da_dtype, int64_t wide_stride, const void *y,
                                         vnDtype y_dtype, int64_t M, int32_t C, int64_t hw, const float *stats,
                                         int32_t relu, float *slab, vnStream stream) {
    VN_CHECK_ARG(slab && M > 0 && rows_ok(C, C) && hw > 0 && M % (2 * hw) == 0 && wide_stride >= 2 * C && (wide_stride & 7) == 0);
    VN_CHECK_ARG(da && y && stats);
    const int rpb = 256 / (C >> 3);
    k_bn_bwd_reduce<false><<<gs_blocks(M, rpb * bwd_rpl(M, rpb), 2048), 256, 0, vn_stream(stream)>>>(da, (int)da_dtype, wide_stride, y, (int)y_dtype,
                                                                                 C, M, C, stats, relu, nullptr, slab, hw, nullptr, nullptr);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_bn_bwd_apply_bev(const void *da, vnDtype da_dtype, int64_t wide_stride, const void *y, vnDtype y_dtype,
                                   int64_t M, int32_t C, int64_t hw, const float *stats, const float *coef, int32_t relu,
                                   void *dy, vnDtype dy_dtype, vnStream stream) {
    VN_CHECK_ARG(M > 0 && rows_ok(C, C) && hw > 0 && M % (2 * hw) == 0 && wide_stride >= 2 * C && (wide_stride & 7) == 0);
    VN_CHECK_ARG(da && y && stats && coef && dy);
    k_bn_bwd_apply<<<gs_blocks(M * (C >> 3), 256 * apply_epl(M * (C >> 3)), 8192), 256, 0, vn_stream(stream)>>>(
        da, (int)da_dtype, wide_stride, y, (int)y_dtype, C, M, C, stats, coef, relu, dy, (int)dy_dtype, C, 0, nullptr, hw);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// vn_bn_bwd_apply at the rows of a site list only (vn_active_sites' (b,d,h,w) list and its device-side count): da, y, dy
// are dense contiguous (B,D,H,W,C) rows.  Rows outside the list are left untouched.
extern "C" int vn_bn_bwd_apply_list(const void *da, vnDtype da_dtype, const void *y, vnDtype y_dtype, int32_t C, int32_t D,
                                    int32_t H, int32_t W, const float *stats, const float *coef, int32_t relu, void *dy,
                                    vnDtype dy_dtype, const int64_t *list, const int32_t *count, int64_t cap,
                                    vnStream stream) {
    VN_CHECK_ARG(rows_ok(C, C) && 256 % (C >> 3) == 0 && D > 0 && H > 0 && W > 0 && cap >= 0);
    if (cap == 0) return VN_OK;
    VN_CHECK_ARG(da && y && stats && coef && dy && list);
    const int rpb = 256 / (C >> 3);
    k_bn_bwd_apply_list<<<gs_blocks(cap, rpb * 2, 4096), 256, 0, vn_stream(stream)>>>(da, (int)da_dtype, y, (int)y_dtype, C, D, H, W,
                                                                                       stats, coef, relu, dy, (int)dy_dtype,
                                                                                       list, count, cap, 0);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

namespace {

// value the activation a = relu(BN(y)) holds at the sites where y = inactive[c] (stored in y's dtype), as stored in a's dtype
__device__ __forceinline__ float inactive_act(const float *stats, const float *inactive, int C, int c, int ydt, int adt, int relu) {
    const float v = as_stored(inactive[c], ydt);
    float z = fmaf(stats[2 * C + c], v - stats[c], stats[3 * C + c]);
    if (relu) z = fmaxf(z, 0.f);
    return as_stored(z, adt);
}

// delta[e][:] = a(site e)[:] - (the inactive sites' activation) for the listed sites: the sparse part of an activation that
// is constant outside the list
__global__ void __launch_bounds__(256) k_act_delta_rows(const void *__restrict__ a, int adt, int C, int D, int H, int W,
                                                        const float *__restrict__ stats, const float *__restrict__ inactive,
                                                        int ydt, int relu, const int64_t *__restrict__ list,
                                                        const int32_t *__restrict__ count, int64_t cap,
                                                        void *__restrict__ out, int odt) {
    VN_PRIO_MAIN();
    const int groups = C >> 3, rpb = 256 / groups;
    const int c = (threadIdx.x % groups) << 3, rr = threadIdx.x / groups;
    if (rr >= rpb) return;
    int64_t n = count ? (int64_t)count[0] : cap;
    if (n > cap) n = cap;
    float cv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cv[j] = inactive_act(stats, inactive, C, c + j, ydt, adt, relu);
    for (int64_t e = (int64_t)blockIdx.x * rpb + rr; e < n; e += (int64_t)gridDim.x * rpb) {
        const int64_t *rc = list + e * 4;
        const int64_t m = ((rc[0] * D + rc[1]) * H + rc[2]) * W + rc[3];
        float v[8];
        load8(a, adt, m * C + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] -= cv[j];
        store8(out, odt, 0, e * C + c, v);
    }
}

// dw[co][ci][kd][kh][kw] += cvec[ci] * Box(kh,kw)[co]: the weight gradient of a 3x3x(kD) convolution (stride 1, padding 1 in
// H/W, none in D) against the CONSTANT part cvec of its input, from the box sums of its output gradient (the slab of
// k_box_partials: see k_box_total).  One workgroup per output channel.
__global__ void __launch_bounds__(256) k_wgrad_const_add(float *__restrict__ dw, const float *__restrict__ slab, int nb, int Co,
                                                         int Ci, int kD, const float *__restrict__ stats,
                                                         const float *__restrict__ inactive, int ydt, int adt, int relu) {
    VN_PRIO_MAIN();
    __shared__ double sums[9];         // Tot, then the eight edges
    __shared__ float box[9];
    const int co = blockIdx.x;
    // (this launch ends the side stream's tail: the nine column sums go to the four waves, shuffle-reduced, one barrier —
    //  they were nine 8-barrier LDS trees in a row)
    for (int k = threadIdx.x >> 6; k < 9; k += 4) {
        const int r0 = k == 0 ? 0 : nb + (k - 1) * BOX_EB, rn = k == 0 ? nb : BOX_EB;
        double a = 0.0;
        for (int r = threadIdx.x & 63; r < rn; r += 64) a += (double)slab[(size_t)(r0 + r) * Co + co];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((threadIdx.x & 63) == 0) sums[k] = a;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int hc = threadIdx.x / 3, wc = threadIdx.x % 3;
        double v = sums[0];
        if (hc == 0) v -= sums[1];
        if (hc == 2) v -= sums[2];
        if (wc == 0) v -= sums[3];
        if (wc == 2) v -= sums[4];
        if (hc == 0 && wc == 0) v += sums[5];
        if (hc == 0 && wc == 2) v += sums[6];
        if (hc == 2 && wc == 0) v += sums[7];
        if (hc == 2 && wc == 2) v += sums[8];
        box[threadIdx.x] = (float)v;
    }
    __syncthreads();
    const int per = kD * 9;
    for (int i = threadIdx.x; i < Ci * per; i += 256) {
        const int ci = i / per, t = i - ci * per;
        const float cv = inactive_act(stats, inactive, Ci, ci, ydt, adt, relu);
        dw[((size_t)co * Ci + ci) * per + t] += cv * box[t % 9];
    }
}

}  // namespace

// ---- BatchNorm backward of a layer whose output is constant outside a site list, from the activation gradient at the
//      listed sites only (see the comment above k_bn_bwd_reduce_list)
extern "C" int64_t vn_bn_bwd_list_slab_rows(int64_t cap, int32_t C) {
    if (cap <= 0 || !rows_ok(C, C) || 256 % (C >> 3) != 0) return 0;
    return gs_blocks(cap, (256 / (C >> 3)) * 4, 1024);
}
extern "C" int vn_bn_bwd_reduce_list(const void *da_rows, vnDtype da_dtype, const void *y, vnDtype y_dtype, int32_t C, int32_t D,
                                     int32_t H, int32_t W, const float *stats, int32_t relu, float *slab, const int64_t *list,
                                     const int32_t *count, int64_t cap, vnStream stream) {
    VN_CHECK_ARG(rows_ok(C, C) && 256 % (C >> 3) == 0 && D > 0 && H > 0 && W > 0 && cap > 0);
    VN_CHECK_ARG(da_rows && y && stats && slab && list);
    k_bn_bwd_reduce_list<<<(unsigned)vn_bn_bwd_list_slab_rows(cap, C), 256, 0, vn_stream(stream)>>>(
        da_rows, (int)da_dtype, y, (int)y_dtype, C, D, H, W, stats, relu, slab, list, count, cap);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
extern "C" int vn_bn_bwd_finalize_list(const float *slab, int64_t slab_rows, int64_t M, int32_t C, const float *gamma,
                                       const float *stats, const float *total, const float *inactive, vnDtype y_dtype,
                                       int32_t relu, float *coef, float *d_gamma, float *d_beta, vnStream stream) {
    VN_CHECK_ARG(slab && gamma && stats && total && inactive && coef && slab_rows > 0 && M > 0 && C > 0);
    k_bn_bwd_finalize_list<<<C, 256, 0, vn_stream(stream)>>>(slab, (int)slab_rows, M, C, gamma, stats, total, inactive, (int)y_dtype,
                                                             relu, coef, d_gamma, d_beta);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
extern "C" int vn_bn_bwd_apply_list_rows(const void *da_rows, vnDtype da_dtype, const void *y, vnDtype y_dtype, int32_t C,
                                         int32_t D, int32_t H, int32_t W, const float *stats, const float *coef, int32_t relu,
                                         void *dy, vnDtype dy_dtype, const int64_t *list, const int32_t *count, int64_t cap,
                                         vnStream stream) {
    VN_CHECK_ARG(rows_ok(C, C) && 256 % (C >> 3) == 0 && D > 0 && H > 0 && W > 0 && cap >= 0);
    if (cap == 0) return VN_OK;
    VN_CHECK_ARG(da_rows && y && stats && coef && dy && list);
    const int rpb = 256 / (C >> 3);
    k_bn_bwd_apply_list<<<gs_blocks(cap, rpb * 2, 4096), 256, 0, vn_stream(stream)>>>(da_rows, (int)da_dtype, y, (int)y_dtype, C, D,
                                                                                       H, W, stats, coef, relu, dy, (int)dy_dtype,
                                                                                       list, count, cap, 1);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
// total[ci] = sum over ALL sites of the data gradient of a 3x3x(kD) convolution (stride 1 and padding 1 in H/W, no padding in
// D) with the torch weight w (Co,Ci,kD,3,3) fp32, from box sums of its dense output gradient dy (B,D,H,W,Co); two launches
extern "C" size_t vn_dgrad_total_workspace_bytes(int32_t Co) { return (size_t)(1024 + 8 * 16) * (size_t)(Co > 0 ? Co : 0) * sizeof(float); }
extern "C" int vn_dgrad_total(const void *dy, vnDtype dy_dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co, int32_t Ci,
                              int32_t kD, const float *w, int32_t dy_sums_to_zero, void *workspace, size_t workspace_bytes,
                              float *total, vnStream stream) {
    VN_CHECK_ARG(dy && w && workspace && total && B > 0 && D > 0 && H > 1 && W > 1 && kD >= 1 && kD <= 3);
    VN_CHECK_ARG(rows_ok(Co, Co) && 256 % (Co >> 3) == 0 && Co <= 256 && Ci > 0);
    if (workspace_bytes < vn_dgrad_total_workspace_bytes(Co)) return VN_EWORKSPACE;
    // dy_sums_to_zero: dy is the output of a train-mode BatchNorm backward — its sum over all sites is zero per channel in
    // exact arithmetic (the output of BatchNorm does not change when a constant is added to its input), so only the eight
    // edge sums are taken (9k rows instead of a pass over the whole tensor)
    const int nb = dy_sums_to_zero ? 0 : 1024;
    float *slab = static_cast<float *>(workspace);
    k_box_partials<<<nb + 8 * BOX_EB, 256, 0, vn_stream(stream)>>>(dy, (int)dy_dtype, B, D, H, W, Co, nb, slab);
    k_box_total<<<Ci, 256, 0, vn_stream(stream)>>>(slab, nb, Co, Ci, kD, w, dy_dtype == VN_BF16 ? 1 : 0, total);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// The box-sum half of vn_dgrad_total on its own (workspace = its slab, reused by vn_wgrad_const_add)
extern "C" int vn_box_col_sums(const void *dy, vnDtype dy_dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co,
                               void *workspace, size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(dy && workspace && B > 0 && D > 0 && H > 1 && W > 1 && rows_ok(Co, Co) && 256 % (Co >> 3) == 0 && Co <= 256);
    if (workspace_bytes < vn_dgrad_total_workspace_bytes(Co)) return VN_EWORKSPACE;
    k_box_partials<<<1024 + 8 * BOX_EB, 256, 0, vn_stream(stream)>>>(dy, (int)dy_dtype, B, D, H, W, Co, 1024,
                                                                     static_cast<float *>(workspace));
    VN_LAUNCH_STATUS();
    return VN_OK;
}
// delta_rows[e][:] = a(listed site e)[:] - cvec, cvec[c] = the value a = relu(BN(y)) holds where y = inactive[c]: the sparse
// part of the first middle layer's activation (model.py:207 + 142 over the sparse grid of model.py:102-106)
extern "C" int vn_act_delta_rows(const void *a, vnDtype a_dtype, int32_t C, int32_t D, int32_t H, int32_t W, const float *stats,
                                 const float *inactive, vnDtype y_dtype, int32_t relu, const int64_t *list,
                                 const int32_t *count, int64_t cap, void *delta_rows, vnDtype delta_dtype, vnStream stream) {
    VN_CHECK_ARG(rows_ok(C, C) && 256 % (C >> 3) == 0 && D > 0 && H > 0 && W > 0 && cap >= 0);
    if (cap == 0) return VN_OK;
    VN_CHECK_ARG(a && stats && inactive && list && delta_rows);
    const int rpb = 256 / (C >> 3);
    k_act_delta_rows<<<gs_blocks(cap, rpb * 2, 4096), 256, 0, vn_stream(stream)>>>(a, (int)a_dtype, C, D, H, W, stats, inactive,
                                                                                    (int)y_dtype, relu, list, count, cap, delta_rows,
                                                                                    (int)delta_dtype);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
// dw (torch (Co,Ci,kD,3,3) fp32) += the weight gradient against the constant part cvec of the input (see vn_act_delta_rows),
// from the box sums of the layer's output gradient left in `workspace` by vn_dgrad_total (dy_sums_to_zero = 0) / vn_box_col_sums
extern "C" int vn_wgrad_const_add(float *dw, const void *workspace, size_t workspace_bytes, int32_t Co, int32_t Ci, int32_t kD,
                                  const float *stats, const float *inactive, vnDtype y_dtype, vnDtype a_dtype, int32_t relu,
                                  vnStream stream) {
    VN_CHECK_ARG(dw && workspace && stats && inactive && Co > 0 && Co <= 256 && Ci > 0 && kD >= 1 && kD <= 3);
    if (workspace_bytes < vn_dgrad_total_workspace_bytes(Co)) return VN_EWORKSPACE;
    k_wgrad_const_add<<<Co, 256, 0, vn_stream(stream)>>>(dw, static_cast<const float *>(workspace), 1024, Co, Ci, kD, stats, inactive,
                                                         (int)y_dtype, (int)a_dtype, relu);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
