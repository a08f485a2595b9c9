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
    if (dtype == VN_F32) {
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
    if (dtype == VN_F32) {
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
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t r = threadIdx.x; r < rows; r += 256) {
        s1 += (double)slab[(r * 2 + 0) * C + c];
        s2 += (double)slab[(r * 2 + 1) * C + c];
    }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)M;
        const double ms = r1[0] / n;
        double var = r2[0] / n - ms * ms;
        if (var < 0.0) var = 0.0;
        const double mean = ms + (shift ? (double)shift[c] : 0.0);
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        if (running_var) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        stats[c] = (float)mean;
        stats[C + c] = invstd;
        stats[2 * C + c] = gamma[c] * invstd;
        stats[3 * C + c] = beta[c];
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
                                                  const uint8_t *__restrict__ flags, const float *__restrict__ inactive) {
    const int groups = C >> 3;
    const int64_t total = M * groups;
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
            v[j] = relu ? fmaxf(z, 0.f) : z;
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
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        s1 += (double)slab[((size_t)r * 2 + 0) * C + c];
        s2 += (double)slab[((size_t)r * 2 + 1) * C + c];
    }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)M;
        const float invstd = stats[C + c];
        const float S = gamma[c] * invstd;
        if (d_gamma) d_gamma[c] = (float)r2[0];
        if (d_beta) d_beta[c] = (float)r1[0];
        coef[c] = S;
        coef[C + c] = -S * invstd * (float)(r2[0] / n);
        coef[2 * C + c] = -S * (float)(r1[0] / n);
    }
}

__global__ void __launch_bounds__(256) k_bn_bwd_finalize(const double *__restrict__ sums, int64_t M, int C, int fold,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ stats, float *__restrict__ coef,
                                                         float *__restrict__ d_gamma, float *__restrict__ d_beta) {
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
                                                           int64_t cap) {
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
        load8(da, dadt, m * C + c, dv);
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
                                                                                          nullptr, nullptr);
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
                                                                                         row_flags, inactive);
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
                                                                                          (int)a_dtype, wide_stride, 0, hw, nullptr, nullptr);
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
                                                                                       list, count, cap);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
