// Fused voxel feature encoder — replaces FeatureLearningNet.forward up to the
// scatter (model.py:93-100) and the two VFELayer.forward calls inside it
// (model.py:74-82), forward and backward, train- or eval-mode BatchNorm1d.
//
//   mask = max_c(x) != 0                                   model.py:95-96
//   h1 = relu(x W1^T + b1); p1 = BN1(h1)                   model.py:75-76 (Linear -> ReLU -> BN)
//   out1 = [p1, max_T p1] * mask                           model.py:77-81
//   h2 = relu(out1 W2^T + b2); p2 = BN2(h2); out2 = [p2, max_T p2] * mask
//   voxelwise = max_T out2                                 model.py:100
//
// HBM-bound by design: the only tensors that touch HBM are the (K,T,7) input,
// the (K,128) output and (backward) one (K,T,16) gradient; the (K,T,32) and
// (K,T,128) intermediates of the reference are recomputed per pass in registers.
// Train-mode BatchNorm statistics are global over all K*T rows (padded slots
// included), so the forward is 3 passes (stats1, stats2, output) and the backward
// 3 passes (BN2 sums, BN1 sums + layer-2 grads, layer-1 grads) with tiny finalize
// kernels between them.
// One wave per voxel, lane = point slot t (T <= 64): the skinny 7->16 and 32->64
// MLPs are fp32 FMAs with wave-uniform weights (scalar loads), i.e. no MFMA.
// Reductions over T (max-pool, argmax, BN sums) go through a per-wave LDS tile
// [T][65] that is then scanned with lane = channel: ~2 LDS ops per element
// instead of a 6-step cross-lane butterfly per channel.
// Per-wave partial sums live in registers across the wave's voxels, are combined
// per workgroup through LDS and written as one slab per workgroup; a reduce kernel
// sums the slabs in a fixed order (deterministic, double precision).
#include "common.h"

namespace {

constexpr int C1 = 16, C2 = 64, CIN = 7;
constexpr int ST1 = 0, ST2 = 4 * C1;            // stats: [mean|invstd|S|beta] per layer
constexpr int STATS_FLOATS = 4 * C1 + 4 * C2;   // 320
constexpr int TS = 65;                          // tile row stride (floats)
// per-wave vector scratch (floats)
constexpr int V_AGG1 = 0, V_U = 16, V_MK = 80, V_R1 = 144, V_G1 = 208, V_R2 = 272, V_G2 = 336, V_AM1 = 400,
              V_DAG1 = 416, V_S = 432, V_SIZE = 512;
constexpr int VFE_BLOCKS_MAX = 1024;
#define VFE_O_UNROLL 4   // the o loops read 16 uniform weights per step from LDS; a full unroll spills
// slab (per workgroup) float counts
constexpr int SLAB_P1 = 2 * C1, SLAB_P2 = 2 * C2, SLAB_B1 = 2 * C2;
constexpr int SLAB_B2 = C2 + C2 * 32 + 64;   // db2 | dW2[64][32] | bn1 sums (32 used, written 64 wide)
constexpr int SLAB_B3 = C1 * CIN + C1;           // dW1 | db1

struct VfeParams {
    const float *w1, *b1, *w2, *b2;
};

__device__ __forceinline__ size_t wave_lds_floats(int T) { return (size_t)(T + 1) * TS + (size_t)T * 16 + V_SIZE; }

struct WaveLds {
    float *tile;   // [T][65]
    float *p1t;    // [T][16]
    float *vec;    // V_SIZE
};

__device__ __forceinline__ WaveLds carve_lds(float *base, int wave, int T) {
    float *p = base + (size_t)wave * wave_lds_floats(T);
    return WaveLds{p, p + (size_t)(T + 1) * TS, p + (size_t)(T + 1) * TS + (size_t)T * 16};
}

// The skinny-MLP weights (2.2k floats) are wave-uniform.  As kernel-argument loads hipcc hoists ~2000 s_loads out
// of the per-voxel loop and spills the SGPRs into VGPR lanes (v_readlane around every FMA: +20k instructions per
// voxel); re-loading them with s_load inside the loop serialises on SMEM latency (4x slower still).  So they live
// in LDS, one copy per workgroup, and are read as broadcast ds_read_b128 (4 weights per LDS instruction).
constexpr int WL_W2A = 0, WL_W2B = 1024, WL_W1 = 2048, WL_B1 = 2160, WL_B2 = 2176, WL_SIZE = 2304;   // floats

// ---- row-lane pieces --------------------------------------------------------------------
__device__ __forceinline__ void load_row(const float *__restrict__ feature, int64_t v, int T, int lane, float x[CIN],
                                         float &m) {
    if (lane < T) {
        const float *f = feature + ((int64_t)v * T + lane) * CIN;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < CIN; ++i) { x[i] = f[i]; mx = fmaxf(mx, x[i]); }
        m = (mx != 0.0f) ? 1.0f : 0.0f;   // model.py:95-96 (NaN != 0 is true in torch as well)
    } else {
#pragma unroll
        for (int i = 0; i < CIN; ++i) x[i] = 0.f;
        m = 0.f;
    }
}

__device__ __forceinline__ void layer1(const VfeParams &P, const float x[CIN], float h1[C1]) {
#pragma unroll
    for (int o = 0; o < C1; ++o) {
        float a = P.b1[o];
#pragma unroll
        for (int i = 0; i < CIN; ++i) a = fmaf(P.w1[o * CIN + i], x[i], a);
        h1[o] = fmaxf(a, 0.f);
    }
}

__device__ __forceinline__ void layer1_lds(const float *__restrict__ wl, const float x[CIN], float h1[C1]) {
#pragma unroll
    for (int o = 0; o < C1; ++o) {
        float a = wl[WL_B1 + o];
#pragma unroll
        for (int i = 0; i < CIN; ++i) a = fmaf(wl[WL_W1 + o * CIN + i], x[i], a);
        h1[o] = fmaxf(a, 0.f);
    }
}

// lane-as-channel scan of tile[t][c] (holding h): max / argmax of p = S*(h-mean)+beta over t < T
__device__ __forceinline__ void scan_max(const float *tile, int T, int c, float mean, float S, float beta, float &mx,
                                         int &amx) {
    mx = -INFINITY;
    amx = 0;
    for (int t = 0; t < T; ++t) {
        const float p = fmaf(S, tile[t * TS + c] - mean, beta);
        if (p > mx) { mx = p; amx = t; }
    }
}

// forward of one voxel up to h2 (row-lane), leaving: tile = h2[t][0..63], vec[V_AGG1], vec[V_MK], p1t = p1*m
// returns per-lane x, m, h1, p1 (unmasked) and h2.  am1 (argmax of p1 over t) is written to vec[V_AM1] when WANT_AM1.
template <bool WANT_AM1>
__device__ __forceinline__ void forward_to_h2(const float *__restrict__ wl, const float *__restrict__ stats,
                                              const WaveLds &L, int T, int lane,
                                              const float x[CIN], float m, float h1[C1], float p1[C1]) {
    const float *w2b_lds = wl + WL_W2B;
    layer1_lds(wl, x, h1);
    // tile <- h1 (16 cols); mask vector
    if (lane < T) {
#pragma unroll
        for (int o = 0; o < C1; ++o) L.tile[lane * TS + o] = h1[o];
        L.vec[V_MK + lane] = m;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < C1) {
        float mx; int amx;
        scan_max(L.tile, T, lane, stats[ST1 + lane], stats[ST1 + 2 * C1 + lane], stats[ST1 + 3 * C1 + lane], mx, amx);
        L.vec[V_AGG1 + lane] = mx;
        if (WANT_AM1) L.vec[V_AM1 + lane] = __int_as_float(amx);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int o = 0; o < C1; ++o)
        p1[o] = fmaf(stats[ST1 + 2 * C1 + o], h1[o] - stats[ST1 + o], stats[ST1 + 3 * C1 + o]);
    // u[o] = sum_i W2[o][16+i] * agg1[i]   (lane = o)
    {
        float u = 0.f;
#pragma unroll
        for (int i = 0; i < C1; ++i) u = fmaf(w2b_lds[i * C2 + lane], L.vec[V_AGG1 + i], u);
        L.vec[V_U + lane] = u;
    }
    if (lane < T) {
#pragma unroll
        for (int i = 0; i < C1; ++i) L.p1t[lane * 16 + i] = p1[i] * m;
    }
    __builtin_amdgcn_wave_barrier();
    // h2[o] = relu(b2[o] + sum_{i<16} W2[o][i]*p1[i]*m + m*u[o]) -> straight into the tile (h1 was consumed
    // by the scan above; this lane's own tile row is written, never held in 64 registers)
    const int trow = (lane < T ? lane : T) * TS;      // lanes >= T dump into the spare row T
#pragma unroll VFE_O_UNROLL
    for (int o = 0; o < C2; ++o) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < C1; ++i) a = fmaf(wl[WL_W2A + o * C1 + i], p1[i], a);
        a = fmaf(m, a + L.vec[V_U + o], wl[WL_B2 + o]);
        L.tile[trow + o] = fmaxf(a, 0.f);
    }
    __builtin_amdgcn_wave_barrier();
}

// combine per-wave lane values across the 4 waves of the workgroup and write the slab
__device__ __forceinline__ void slab_write(float *red /*[4][n]*/, const float *vals, int nvals_per_lane, int lane,
                                           int wave, float *slab) {
    // red layout: [wave][j*64 + lane]
    const int n = nvals_per_lane * 64;
    for (int j = 0; j < nvals_per_lane; ++j) red[wave * n + j * 64 + lane] = vals[j];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) slab[i] = red[i] + red[n + i] + red[2 * n + i] + red[3 * n + i];
}

__device__ __forceinline__ void load_weights_lds(const VfeParams &P, float *wl) {
    for (int idx = threadIdx.x; idx < C1 * C2; idx += 256) {
        const int o = idx / C1, i = idx - o * C1;
        wl[WL_W2A + idx] = P.w2[o * 32 + i];                 // first half of W2, [o][i]
        const int i2 = idx / C2, o2 = idx - i2 * C2;
        wl[WL_W2B + idx] = P.w2[o2 * 32 + 16 + i2];          // second half transposed: [i][o] = W2[o][16+i]
    }
    for (int idx = threadIdx.x; idx < C1 * CIN; idx += 256) wl[WL_W1 + idx] = P.w1[idx];
    if (threadIdx.x < C1) wl[WL_B1 + threadIdx.x] = P.b1[threadIdx.x];
    if (threadIdx.x < C2) wl[WL_B2 + threadIdx.x] = P.b2[threadIdx.x];
    __syncthreads();
}

// ---- forward passes ---------------------------------------------------------------------
// pass 1: sums of h1 ; slab[b] = [sum(16) | sumsq(16)]
__global__ void __launch_bounds__(256) k_vfe_p1(const float *__restrict__ feature, int64_t K, int T, VfeParams P,
                                                float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = smem + (size_t)wave * T * TS;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < K; v += (int64_t)gridDim.x * 4) {
        asm volatile("" ::: "memory");   // uniform operands are re-read from LDS per voxel, not hoisted into 2000 registers
        float x[CIN], m, h1[C1];
        load_row(feature, v, T, lane, x, m);
        layer1(P, x, h1);
        if (lane < T) {
#pragma unroll
            for (int o = 0; o < C1; ++o) tile[lane * TS + o] = h1[o];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < C1) {
            for (int t = 0; t < T; ++t) { const float h = tile[t * TS + lane]; s1 += h; s2 += h * h; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // pack: lanes 0..15 sum, lanes 16..31 sumsq
    const float sq = __shfl(s2, lane & 15, 64);
    float vals[1] = {lane < C1 ? s1 : (lane < 2 * C1 ? sq : 0.f)};
    slab_write(smem, vals, 1, lane, wave, slabs + (size_t)blockIdx.x * 64);
}

// pass 2: sums of h2 ; slab[b] = [sum(64) | sumsq(64)]
__global__ void __launch_bounds__(256) k_vfe_p2(const float *__restrict__ feature, int64_t K, int T, VfeParams P,
                                                const float *__restrict__ stats, float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *wl = smem;
    load_weights_lds(P, wl);
    const WaveLds L = carve_lds(smem + WL_SIZE, wave, T);
    float s1 = 0.f, s2 = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < K; v += (int64_t)gridDim.x * 4) {
        asm volatile("" ::: "memory");   // uniform operands are re-read from LDS per voxel, not hoisted into 2000 registers
        float x[CIN], m, h1[C1], p1[C1];
        load_row(feature, v, T, lane, x, m);
        forward_to_h2<false>(wl, stats, L, T, lane, x, m, h1, p1);
        for (int t = 0; t < T; ++t) { const float h = L.tile[t * TS + lane]; s1 += h; s2 += h * h; }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float vals[2] = {s1, s2};
    slab_write(smem, vals, 2, lane, wave, slabs + (size_t)blockIdx.x * 128);
}

// pass 3: voxelwise output (K,128)
__global__ void __launch_bounds__(256) k_vfe_p3(const float *__restrict__ feature, int64_t K, int T, VfeParams P,
                                                const float *__restrict__ stats, float *__restrict__ voxelwise) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *wl = smem;
    load_weights_lds(P, wl);
    const WaveLds L = carve_lds(smem + WL_SIZE, wave, T);
    const float mean2 = stats[ST2 + lane], S2 = stats[ST2 + 2 * C2 + lane], be2 = stats[ST2 + 3 * C2 + lane];
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < K; v += (int64_t)gridDim.x * 4) {
        asm volatile("" ::: "memory");   // uniform operands are re-read from LDS per voxel, not hoisted into 2000 registers
        float x[CIN], m, h1[C1], p1[C1];
        load_row(feature, v, T, lane, x, m);
        forward_to_h2<false>(wl, stats, L, T, lane, x, m, h1, p1);
        // lane = channel: agg2 = max_t p2 ; vw_lo = max_t p2*m ; vw_hi = max_t agg2*m
        float agg = -INFINITY, vlo = -INFINITY;
        float anym = 0.f, allm = 1.f;
        for (int t = 0; t < T; ++t) {
            const float p = fmaf(S2, L.tile[t * TS + lane] - mean2, be2);
            const float mk = L.vec[V_MK + t];
            agg = fmaxf(agg, p);
            vlo = fmaxf(vlo, p * mk);
            anym = fmaxf(anym, mk);
            allm = fminf(allm, mk);
        }
        float vhi = agg * anym;                       // all masks equal -> agg*m
        if (anym != allm) vhi = fmaxf(agg, 0.f);      // both 0 and 1 present
        voxelwise[v * 128 + lane] = vlo;
        voxelwise[v * 128 + 64 + lane] = vhi;
        __builtin_amdgcn_wave_barrier();
    }
}

// block-wide sum of slab columns: pair (c, C + c) of every slab, in double (fixed order -> deterministic)
__device__ __forceinline__ void slab_pair_sum(const float *__restrict__ slabs, int nslabs, int stride, int off, int C,
                                              int c, double &o1, double &o2) {
    __shared__ double r1[256], r2[256];
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nslabs; b += 256) {
        s1 += slabs[(size_t)b * stride + off + c];
        s2 += slabs[(size_t)b * stride + off + C + c];
    }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    o1 = r1[0];
    o2 = r2[0];
}

// slabs -> stats (train) or running stats -> stats (eval); one workgroup per channel
__global__ void __launch_bounds__(256) k_vfe_finalize(const float *__restrict__ slabs, int nslabs, int slab_stride, int C,
                                                      int64_t rows,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      float *running_mean, float *running_var, int training,
                                                      float momentum, float eps, float *__restrict__ st) {
    const int c = blockIdx.x;
    double mean, var;
    if (training) {
        double s1, s2;
        slab_pair_sum(slabs, nslabs, slab_stride, 0, C, c, s1, s2);
        const double n = (double)rows;
        mean = s1 / n;
        var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        if (threadIdx.x == 0) {
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    if (threadIdx.x == 0) {
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        st[c] = (float)mean;
        st[C + c] = invstd;
        st[2 * C + c] = gamma[c] * invstd;
        st[3 * C + c] = beta[c];
    }
}

// ---- backward passes --------------------------------------------------------------------
// channel-lane analysis of layer-2 outputs for one voxel: impulses of d_p2
//   r1 = argmax_t p2*m (first), g1 = dvw[c]*m[r1] ; r2 = argmax_t p2, g2 = dvw[64+c]*m[a'], a' = argmax_t agg2*m
__device__ __forceinline__ void impulses(const WaveLds &L, int T, int lane, float mean2, float S2, float be2,
                                         float dlo, float dhi, int &r1, float &g1, int &r2, float &g2, float &xh1,
                                         float &xh2, float inv2) {
    float agg = -INFINITY, vlo = -INFINITY;
    r1 = 0; r2 = 0;
    float h_r1 = 0.f, h_r2 = 0.f;
    for (int t = 0; t < T; ++t) {
        const float h = L.tile[t * TS + lane];
        const float p = fmaf(S2, h - mean2, be2);
        const float pm = p * L.vec[V_MK + t];
        if (p > agg) { agg = p; r2 = t; h_r2 = h; }
        if (pm > vlo) { vlo = pm; r1 = t; h_r1 = h; }
    }
    // a' = first t maximising agg*m[t]
    int ap = 0;
    float best = -INFINITY;
    for (int t = 0; t < T; ++t) {
        const float q = agg * L.vec[V_MK + t];
        if (q > best) { best = q; ap = t; }
    }
    g1 = dlo * L.vec[V_MK + r1];
    g2 = dhi * L.vec[V_MK + ap];
    xh1 = (h_r1 - mean2) * inv2;
    xh2 = (h_r2 - mean2) * inv2;
}

// backward pass 1: BN2 sums ; slab = [sum d_p2 (64) | sum d_p2*xhat2 (64)]
__global__ void __launch_bounds__(256) k_vfe_b1(const float *__restrict__ feature, int64_t K, int T, VfeParams P,
                                                const float *__restrict__ stats, const float *__restrict__ dvw,
                                                float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *wl = smem;
    load_weights_lds(P, wl);
    const WaveLds L = carve_lds(smem + WL_SIZE, wave, T);
    const float mean2 = stats[ST2 + lane], inv2 = stats[ST2 + C2 + lane], S2 = stats[ST2 + 2 * C2 + lane],
                be2 = stats[ST2 + 3 * C2 + lane];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < K; v += (int64_t)gridDim.x * 4) {
        asm volatile("" ::: "memory");   // uniform operands are re-read from LDS per voxel, not hoisted into 2000 registers
        float x[CIN], m, h1[C1], p1[C1];
        load_row(feature, v, T, lane, x, m);
        forward_to_h2<false>(wl, stats, L, T, lane, x, m, h1, p1);
        int r1, r2; float g1, g2, xh1, xh2;
        impulses(L, T, lane, mean2, S2, be2, dvw[v * 128 + lane], dvw[v * 128 + 64 + lane], r1, g1, r2, g2, xh1, xh2,
                 inv2);
        s1 += g1 + g2;
        s2 += g1 * xh1 + g2 * xh2;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float vals[2] = {s1, s2};
    slab_write(smem, vals, 2, lane, wave, slabs + (size_t)blockIdx.x * 128);
}

// BN backward finalize from slabs: coef = [c0|c1|c2], d_gamma, d_beta ; one workgroup per channel
__global__ void __launch_bounds__(256) k_vfe_bn_bwd_finalize(const float *__restrict__ slabs, int nslabs, int slab_stride,
                                                             int slab_off, int C, int64_t rows,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ st, float *__restrict__ coef,
                                                             float *__restrict__ d_gamma, float *__restrict__ d_beta) {
    const int c = blockIdx.x;
    double s1, s2;
    slab_pair_sum(slabs, nslabs, slab_stride, slab_off, C, c, s1, s2);
    if (threadIdx.x == 0) {
        const double n = (double)rows;
        const float invstd = st[C + c];
        const float S = gamma[c] * invstd;
        coef[c] = S;
        coef[C + c] = -S * invstd * (float)(s2 / n);
        coef[2 * C + c] = -S * (float)(s1 / n);
        d_gamma[c] = (float)s2;
        d_beta[c] = (float)s1;
    }
}

// backward pass 2: layer-2 parameter grads, d_p1 rows -> workspace, BN1 sums
//   slab = [db2 (64) | dW2 (64*32) | sum d_p1 (16) | sum d_p1*xhat1 (16)]
__global__ void __launch_bounds__(256) k_vfe_b2(const float *__restrict__ feature, int64_t K, int T, VfeParams P,
                                                const float *__restrict__ stats_g, const float *__restrict__ dvw,
                                                const float *__restrict__ coef2_g, float *__restrict__ dp1_ws,
                                                float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *wl = smem;
    load_weights_lds(P, wl);
    // this kernel also keeps the BN statistics and the BN2 backward coefficients in LDS (512 more uniform floats)
    float *st_l = smem + WL_SIZE, *cf_l = st_l + STATS_FLOATS;
    for (int i = threadIdx.x; i < STATS_FLOATS; i += 256) st_l[i] = stats_g[i];
    for (int i = threadIdx.x; i < 3 * C2; i += 256) cf_l[i] = coef2_g[i];
    __syncthreads();
    const float *stats = st_l, *coef2 = cf_l;
    const WaveLds L = carve_lds(smem + WL_SIZE + 512, wave, T);
    const float mean2 = stats[ST2 + lane], inv2 = stats[ST2 + C2 + lane], S2 = stats[ST2 + 2 * C2 + lane],
                be2 = stats[ST2 + 3 * C2 + lane];
    float db2 = 0.f, dw2a[C1], dw2b[C1], bn1 = 0.f;
#pragma unroll
    for (int i = 0; i < C1; ++i) { dw2a[i] = 0.f; dw2b[i] = 0.f; }
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < K; v += (int64_t)gridDim.x * 4) {
        asm volatile("" ::: "memory");   // uniform operands are re-read from LDS per voxel, not hoisted into 2000 registers
        float x[CIN], m, h1[C1], p1[C1];
        load_row(feature, v, T, lane, x, m);
        forward_to_h2<true>(wl, stats, L, T, lane, x, m, h1, p1);
        {
            int r1, r2; float g1, g2, xh1, xh2;
            impulses(L, T, lane, mean2, S2, be2, dvw[v * 128 + lane], dvw[v * 128 + 64 + lane], r1, g1, r2, g2, xh1,
                     xh2, inv2);
            L.vec[V_R1 + lane] = __int_as_float(r1);
            L.vec[V_G1 + lane] = g1;
            L.vec[V_R2 + lane] = __int_as_float(r2);
            L.vec[V_G2 + lane] = g2;
        }
        __builtin_amdgcn_wave_barrier();
        // row-lane: d_pre2[o] = (h2>0) * (c0*d_p2 + c1*(h2-mean) + c2), streamed through the tile IN PLACE
        // (h2[o] read, d_pre2[o] written back to the same slot) while d_p1m[i] = sum_o d_pre2[o]*W2[o][i] accumulates
        float dp1[C1];
#pragma unroll
        for (int i = 0; i < C1; ++i) dp1[i] = 0.f;
        const int trow = (lane < T ? lane : T) * TS;
#pragma unroll VFE_O_UNROLL
        for (int o = 0; o < C2; ++o) {
            const float h = L.tile[trow + o];
            float dp = 0.f;
            if (__float_as_int(L.vec[V_R1 + o]) == lane) dp += L.vec[V_G1 + o];
            if (__float_as_int(L.vec[V_R2 + o]) == lane) dp += L.vec[V_G2 + o];
            const float dh = fmaf(coef2[o], dp, fmaf(coef2[C2 + o], h - stats[ST2 + o], coef2[2 * C2 + o]));
            const float d = (h > 0.f && lane < T) ? dh : 0.f;
            L.tile[trow + o] = d;
#pragma unroll
            for (int i = 0; i < C1; ++i) dp1[i] = fmaf(d, wl[WL_W2A + o * C1 + i], dp1[i]);
        }
        __builtin_amdgcn_wave_barrier();
        // lane = o: db2, s[o] = sum_t m_t dpre, dW2a[o][i] += sum_t dpre[t][o]*p1m[t][i]
        float s = 0.f;
        for (int t = 0; t < T; ++t) {
            const float d = L.tile[t * TS + lane];
            db2 += d;
            s = fmaf(L.vec[V_MK + t], d, s);
#pragma unroll
            for (int i = 0; i < C1; ++i) dw2a[i] = fmaf(d, L.p1t[t * 16 + i], dw2a[i]);
        }
#pragma unroll
        for (int i = 0; i < C1; ++i) dw2b[i] = fmaf(L.vec[V_AGG1 + i], s, dw2b[i]);
        L.vec[V_S + lane] = s;
        __builtin_amdgcn_wave_barrier();
        // lane = i' < 16: d_agg1[i'] = sum_o W2[o][16+i'] * s[o]
        if (lane < C1) {
            float da = 0.f;
            for (int o = 0; o < C2; ++o) da = fmaf(wl[WL_W2B + lane * C2 + o], L.vec[V_S + o], da);
            L.vec[V_DAG1 + lane] = da;
        }
        __builtin_amdgcn_wave_barrier();
        // row-lane: d_p1[i] = d_p1m[i]*m + [t == am1[i]] * d_agg1[i] ; store ; BN1 sums through the tile
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            float d = dp1[i] * m;
            if (__float_as_int(L.vec[V_AM1 + i]) == lane) d += L.vec[V_DAG1 + i];
            dp1[i] = d;
        }
        if (lane < T) {
            float *dst = dp1_ws + ((int64_t)v * T + lane) * C1;
#pragma unroll
            for (int i = 0; i < C1; i += 4)
                *reinterpret_cast<float4 *>(dst + i) = make_float4(dp1[i], dp1[i + 1], dp1[i + 2], dp1[i + 3]);
#pragma unroll
            for (int i = 0; i < C1; ++i) {
                L.tile[lane * TS + i] = dp1[i];
                L.tile[lane * TS + C1 + i] = dp1[i] * ((h1[i] - stats[ST1 + i]) * stats[ST1 + C1 + i]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 2 * C1) {
            for (int t = 0; t < T; ++t) bn1 += L.tile[t * TS + lane];
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // slab: [db2 | dW2 | bn1]: write through LDS in chunks
    float *slab = slabs + (size_t)blockIdx.x * SLAB_B2;
    {
        float vals[1] = {db2};
        slab_write(smem, vals, 1, lane, wave, slab);   // db2[o] at slab[o]
        __syncthreads();
    }
    {
        // dW2[o][i] = dw2a[i], dW2[o][16+i] = dw2b[i] ; combined through LDS as [j][lane] then transposed on store
        const int n = 32 * 64;
        for (int j = 0; j < C1; ++j) {
            smem[wave * n + j * 64 + lane] = dw2a[j];
            smem[wave * n + (C1 + j) * 64 + lane] = dw2b[j];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const int j = i >> 6, o = i & 63;
            slab[C2 + o * 32 + j] = smem[i] + smem[n + i] + smem[2 * n + i] + smem[3 * n + i];
        }
        __syncthreads();
    }
    {
        float vals[1] = {lane < 2 * C1 ? bn1 : 0.f};
        slab_write(smem, vals, 1, lane, wave, slab + C2 + C2 * 32);   // 64 wide, first 32 meaningful
    }
}

// backward pass 3: layer-1 parameter grads ; slab = [dW1 (16*7) | db1 (16)]
__global__ void __launch_bounds__(256) k_vfe_b3(const float *__restrict__ feature, int64_t K, int T, VfeParams P,
                                                const float *__restrict__ stats, const float *__restrict__ coef1,
                                                const float *__restrict__ dp1_ws, float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = smem + (size_t)wave * T * TS;
    // lane handles entries e0 = lane, e1 = lane + 64 of [dW1 (112) | db1 (16)]
    const int e0 = lane, e1 = lane + 64;
    const int o0 = e0 / CIN, i0 = e0 - o0 * CIN;                 // e0 < 64 < 112 always a dW1 entry
    const bool w1e = e1 < C1 * CIN;
    const int o1 = w1e ? e1 / CIN : e1 - C1 * CIN, i1 = w1e ? e1 - o1 * CIN : 0;
    float a0 = 0.f, a1 = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < K; v += (int64_t)gridDim.x * 4) {
        asm volatile("" ::: "memory");   // uniform operands are re-read from LDS per voxel, not hoisted into 2000 registers
        float x[CIN], m, h1[C1];
        load_row(feature, v, T, lane, x, m);
        layer1(P, x, h1);
        if (lane < T) {
            const float *src = dp1_ws + ((int64_t)v * T + lane) * C1;
#pragma unroll
            for (int i = 0; i < C1; i += 4) {
                const float4 d = *reinterpret_cast<const float4 *>(src + i);
                const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = i + j;
                    const float dh = fmaf(coef1[o], dd[j], fmaf(coef1[C1 + o], h1[o] - stats[ST1 + o], coef1[2 * C1 + o]));
                    tile[lane * TS + o] = h1[o] > 0.f ? dh : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < CIN; ++i) tile[lane * TS + C1 + i] = x[i];
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = 0; t < T; ++t) {
            a0 = fmaf(tile[t * TS + o0], tile[t * TS + C1 + i0], a0);
            a1 = w1e ? fmaf(tile[t * TS + o1], tile[t * TS + C1 + i1], a1) : a1 + tile[t * TS + o1];
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float vals[2] = {a0, a1};
    slab_write(smem, vals, 2, lane, wave, slabs + (size_t)blockIdx.x * 128);
}

// out[i] = sum_b slabs[b*stride + off + i]  (double accumulation, fixed order); one wave per output element
__global__ void __launch_bounds__(256) k_vfe_reduce(const float *__restrict__ slabs, int nslabs, int stride, int off,
                                                    int n, float *__restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    double s = 0.0;
    for (int b = lane; b < nslabs; b += 64) s += slabs[(size_t)b * stride + off + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[i] = (float)s;
}

struct Plan {
    int blocks;
    size_t lds_small, lds_full;
    size_t off_slabs, off_coef, off_dp1, bytes;
};

Plan make_plan(int64_t K, int T) {
    Plan p{};
    int64_t b = vn_ceil_div(K, 4);
    if (b < 1) b = 1;
    if (b > VFE_BLOCKS_MAX) b = VFE_BLOCKS_MAX;
    p.blocks = (int)b;
    p.lds_small = (size_t)4 * T * TS * sizeof(float);
    const size_t per_wave = ((size_t)(T + 1) * TS + (size_t)T * 16 + V_SIZE) * sizeof(float);
    size_t full = (size_t)(WL_SIZE + 512) * sizeof(float) + 4 * per_wave;
    const size_t red = (size_t)4 * 32 * 64 * sizeof(float);   // slab combine area of pass b2
    if (full < red) full = red;
    p.lds_full = full;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += vn_align(bytes); return r; };
    p.off_slabs = take((size_t)VFE_BLOCKS_MAX * SLAB_B2 * sizeof(float));
    p.off_coef = take((size_t)(3 * C1 + 3 * C2) * sizeof(float));
    p.off_dp1 = take((size_t)(K > 0 ? K : 1) * T * C1 * sizeof(float));
    p.bytes = off;
    return p;
}

}  // namespace

extern "C" size_t vn_vfe_workspace_bytes(int64_t K, int32_t T) {
    if (K < 0 || T <= 0 || T > 64) return 0;
    return make_plan(K, T).bytes;
}

extern "C" int vn_vfe_fwd(const float *feature, int64_t K, int32_t T, const vnVfeWeights *w, int32_t training,
                          float momentum, float eps, float *voxelwise, float *stats, void *workspace,
                          size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(w && stats && workspace && K >= 0 && T > 0 && T <= 64);
    VN_CHECK_ARG(w->w1 && w->b1 && w->g1 && w->be1 && w->rm1 && w->rv1 && w->w2 && w->b2 && w->g2 && w->be2 && w->rm2 &&
                 w->rv2);
    const Plan pl = make_plan(K, T);
    if (workspace_bytes < pl.bytes) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    float *slabs = reinterpret_cast<float *>(static_cast<char *>(workspace) + pl.off_slabs);
    const VfeParams P{w->w1, w->b1, w->w2, w->b2};
    static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vfe_p2),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t a2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vfe_p3),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t a3 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vfe_p1),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (a1 != hipSuccess || a2 != hipSuccess || a3 != hipSuccess) return (int)(a1 != hipSuccess ? a1 : a2 != hipSuccess ? a2 : a3);
    const int64_t rows = K * T;
    if (training) {
        VN_CHECK_ARG(K > 0);
        k_vfe_p1<<<pl.blocks, 256, pl.lds_small, st>>>(feature, K, T, P, slabs);
        VN_LAUNCH_STATUS();
        k_vfe_finalize<<<C1, 256, 0, st>>>(slabs, pl.blocks, 64, C1, rows, w->g1, w->be1, w->rm1, w->rv1, 1, momentum, eps,
                                         stats + ST1);
        VN_LAUNCH_STATUS();
        k_vfe_p2<<<pl.blocks, 256, pl.lds_full, st>>>(feature, K, T, P, stats, slabs);
        VN_LAUNCH_STATUS();
        k_vfe_finalize<<<C2, 256, 0, st>>>(slabs, pl.blocks, 128, C2, rows, w->g2, w->be2, w->rm2, w->rv2, 1, momentum, eps,
                                         stats + ST2);
        VN_LAUNCH_STATUS();
    } else {
        k_vfe_finalize<<<C1, 256, 0, st>>>(nullptr, 0, 0, C1, rows, w->g1, w->be1, w->rm1, w->rv1, 0, momentum, eps, stats + ST1);
        VN_LAUNCH_STATUS();
        k_vfe_finalize<<<C2, 256, 0, st>>>(nullptr, 0, 0, C2, rows, w->g2, w->be2, w->rm2, w->rv2, 0, momentum, eps, stats + ST2);
        VN_LAUNCH_STATUS();
    }
    if (K == 0) return VN_OK;
    VN_CHECK_ARG(feature && voxelwise);
    k_vfe_p3<<<pl.blocks, 256, pl.lds_full, st>>>(feature, K, T, P, stats, voxelwise);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_vfe_bwd(const float *feature, int64_t K, int32_t T, const vnVfeWeights *w, const float *stats,
                          const float *d_voxelwise, const vnVfeGrads *g, void *workspace, size_t workspace_bytes,
                          vnStream stream) {
    VN_CHECK_ARG(feature && w && stats && d_voxelwise && g && workspace && K > 0 && T > 0 && T <= 64);
    VN_CHECK_ARG(g->dw1 && g->db1 && g->dg1 && g->dbe1 && g->dw2 && g->db2 && g->dg2 && g->dbe2);
    const Plan pl = make_plan(K, T);
    if (workspace_bytes < pl.bytes) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    char *ws = static_cast<char *>(workspace);
    float *slabs = reinterpret_cast<float *>(ws + pl.off_slabs);
    float *coef1 = reinterpret_cast<float *>(ws + pl.off_coef);
    float *coef2 = coef1 + 3 * C1;
    float *dp1 = reinterpret_cast<float *>(ws + pl.off_dp1);
    const VfeParams P{w->w1, w->b1, w->w2, w->b2};
    static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vfe_b1),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t a2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vfe_b2),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static const hipError_t a3 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vfe_b3),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (a1 != hipSuccess || a2 != hipSuccess || a3 != hipSuccess) return (int)(a1 != hipSuccess ? a1 : a2 != hipSuccess ? a2 : a3);
    const int64_t rows = K * T;
    k_vfe_b1<<<pl.blocks, 256, pl.lds_full, st>>>(feature, K, T, P, stats, d_voxelwise, slabs);
    VN_LAUNCH_STATUS();
    k_vfe_bn_bwd_finalize<<<C2, 256, 0, st>>>(slabs, pl.blocks, 128, 0, C2, rows, w->g2, stats + ST2, coef2, g->dg2, g->dbe2);
    VN_LAUNCH_STATUS();
    k_vfe_b2<<<pl.blocks, 256, pl.lds_full, st>>>(feature, K, T, P, stats, d_voxelwise, coef2, dp1, slabs);
    VN_LAUNCH_STATUS();
    k_vfe_reduce<<<C2 / 4, 256, 0, st>>>(slabs, pl.blocks, SLAB_B2, 0, C2, g->db2);
    VN_LAUNCH_STATUS();
    k_vfe_reduce<<<C2 * 32 / 4, 256, 0, st>>>(slabs, pl.blocks, SLAB_B2, C2, C2 * 32, g->dw2);
    VN_LAUNCH_STATUS();
    k_vfe_bn_bwd_finalize<<<C1, 256, 0, st>>>(slabs, pl.blocks, SLAB_B2, C2 + C2 * 32, C1, rows, w->g1, stats + ST1, coef1,
                                            g->dg1, g->dbe1);
    VN_LAUNCH_STATUS();
    k_vfe_b3<<<pl.blocks, 256, pl.lds_small, st>>>(feature, K, T, P, stats, coef1, dp1, slabs);
    VN_LAUNCH_STATUS();
    k_vfe_reduce<<<C1 * CIN / 4, 256, 0, st>>>(slabs, pl.blocks, 128, 0, C1 * CIN, g->dw1);
    VN_LAUNCH_STATUS();
    k_vfe_reduce<<<C1 / 4, 256, 0, st>>>(slabs, pl.blocks, 128, C1 * CIN, C1, g->db1);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
