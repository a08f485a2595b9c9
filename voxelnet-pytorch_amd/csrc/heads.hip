// The two 1x1 heads of the RPN (model.py:276-281: Conv2d(768, 2) + sigmoid, Conv2d(768, 14)) as ONE 16-channel product over
// the 768-channel concat, forward and data gradient, bf16.  Through k_gather_gemm these two launches were 44 + 4 us and
// 66 us for 108 MB each way (N = 16 padded to a 64-column tile, K steps of 64 with one stage of lookahead): they are
// streaming passes, so here
//   forward:        a wave keeps the whole 16 x 768 weight matrix as 24 MFMA B fragments in registers and streams 16-row
//                   groups: 24 independent 16-B loads per lane in flight, 24 v_mfma_f32_16x16x32_bf16, then bias, sigmoid on
//                   the two score channels and the NCHW stores (4 consecutive sites per lane) — no (M,16) intermediate;
//   data gradient:  K = 16: v_mfma_f32_16x16x16_bf16 with the weights as the A operand (rows = channels), rows permuted so
//                   that a lane's two tiles hold 8 consecutive channels: one 16-B store per lane and tile pair.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__global__ void __launch_bounds__(256) k_heads_fwd(const bf16_t *__restrict__ cat, int64_t cat_stride,
                                                   const bf16_t *__restrict__ w /* [16][768] */, const float *__restrict__ bias,
                                                   int B, int64_t S, float *__restrict__ prob, float *__restrict__ reg) {
    VN_PRIO_MAIN();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8_t Bf[24];
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) Bf[ks] = *reinterpret_cast<const bf16x8_t *>(w + fr * 768 + ks * 32 + fq * 8);
    const float bv = bias ? bias[fr] : 0.f;
    const int64_t M = (int64_t)B * S, ngroups = (M + 15) >> 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < ngroups; g += (int64_t)gridDim.x * 4) {
        const int64_t m0 = g << 4;
        int64_t row = m0 + fr;
        if (row >= M) row = M - 1;
        const bf16_t *ap = cat + row * cat_stride + fq * 8;
        bf16x8_t A[24];
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) A[ks] = *reinterpret_cast<const bf16x8_t *>(ap + ks * 32);
        // (all 24 loads are issued before the first MFMA: left alone, the scheduler sinks every load to its use and the wave
        //  makes 24 sequential round trips to HBM — measured 46 us for this launch instead of 25)
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks], Bf[ks], acc, 0, 0, 0);
        // lane: sites m0 + fq*4 + e (e = 0..3), channel fr
        const int64_t m = m0 + fq * 4;
        if (m >= M) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = acc[e] + bv;
            v[e] = fr < 2 ? 1.0f / (1.0f + expf(-x)) : x;
        }
        const int64_t b = m / S, s = m - b * S;
        float *dst = fr < 2 ? prob + (b * 2 + fr) * S + s : reg + (b * 14 + (fr - 2)) * S + s;
        if ((S & 3) == 0 && m + 3 < M) {
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int e = 0; e < 4; ++e) {
                const int64_t me = m + e;
                if (me >= M) break;
                const int64_t be = me / S, se = me - be * S;
                (fr < 2 ? prob + (be * 2 + fr) * S + se : reg + (be * 14 + (fr - 2)) * S + se)[0] = v[e];
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_heads_dgrad(const bf16_t *__restrict__ drows /* [M][16] */, int64_t drows_stride,
                                                     const bf16_t *__restrict__ wd /* [768][16] */, bf16_t *__restrict__ dcat,
                                                     int64_t dcat_stride, int64_t M) {
    VN_PRIO_MAIN();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // MFMA row r = 4g + e of tile h of pair p <-> channel 32p + 8g + 4h + e: a lane (rows 4*fq .. +3 of both tiles) ends up
    // with the 8 consecutive channels 32p + 8*fq .. +7
    s16x4_t Wf[48];
#pragma unroll
    for (int p = 0; p < 24; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 32 * p + 8 * (fr >> 2) + 4 * h + (fr & 3);
            Wf[2 * p + h] = *reinterpret_cast<const s16x4_t *>(wd + c * 16 + fq * 4);
        }
    const int64_t ngroups = (M + 15) >> 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < ngroups; g += (int64_t)gridDim.x * 4) {
        const int64_t m = (g << 4) + fr;                   // this lane's site (MFMA column)
        const bool ok = m < M;
        const s16x4_t G = *reinterpret_cast<const s16x4_t *>(drows + (ok ? m : M - 1) * drows_stride + fq * 4);
        bf16_t *orow = dcat + m * dcat_stride + 8 * fq;
#pragma unroll
        for (int p = 0; p < 24; ++p) {
            const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Wf[2 * p], G, z, 0, 0, 0);
            const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Wf[2 * p + 1], G, z, 0, 0, 0);
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)d0[e]; o[4 + e] = (bf16_t)d1[e]; }
            if (ok) *reinterpret_cast<bf16x8_t *>(orow + 32 * p) = o;
        }
    }
}

// The same data gradient WITHOUT operand rounding: d_rows (M,16) and the weights (16,768) in fp32, fp32 FMAs (the 1.7 GFLOP
// of this product hide behind the 108-216 MB it writes).  The logit gradient is almost the same number at every negative
// anchor and each deconv's train-mode BatchNorm backward removes most of what the concat gradient has in common over the
// sites: the bf16 roundings of d_rows and of the weights are a 2^-9 perturbation of that common part, i.e. a 10-30 %
// perturbation of what is left.  A thread owns 8 consecutive channels (its 16 x 8 weights in
// registers) and walks rows; the 96 threads of a row read the same 64 B of d_rows (L1 broadcast).
template <bool OUT_F32>
__global__ void __launch_bounds__(192) k_heads_dgrad_f32(const float *__restrict__ drows, int64_t drows_stride,
                                                         const float *__restrict__ w /* [16][768] */, void *__restrict__ dcat,
                                                         int64_t dcat_stride, int64_t M) {
    VN_PRIO_MAIN();
    const int cg = threadIdx.x % 96, rr = threadIdx.x / 96;      // channel group (8 channels), row slot 0/1
    float W[16][8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 a = *reinterpret_cast<const float4 *>(w + j * 768 + cg * 8);
        const float4 b = *reinterpret_cast<const float4 *>(w + j * 768 + cg * 8 + 4);
        W[j][0] = a.x; W[j][1] = a.y; W[j][2] = a.z; W[j][3] = a.w; W[j][4] = b.x; W[j][5] = b.y; W[j][6] = b.z; W[j][7] = b.w;
    }
    for (int64_t m = (int64_t)blockIdx.x * 2 + rr; m < M; m += (int64_t)gridDim.x * 2) {
        float g[16];
        const float4 *gp = reinterpret_cast<const float4 *>(drows + m * drows_stride);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t = gp[q]; g[4 * q] = t.x; g[4 * q + 1] = t.y; g[4 * q + 2] = t.z; g[4 * q + 3] = t.w; }
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(g[j], W[j][e], o[e]);
        if constexpr (OUT_F32) {
            float *d = static_cast<float *>(dcat) + m * dcat_stride + cg * 8;
            *reinterpret_cast<float4 *>(d) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4 *>(d + 4) = make_float4(o[4], o[5], o[6], o[7]);
        } else {
            bf16x8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)o[e];
            *reinterpret_cast<bf16x8_t *>(static_cast<bf16_t *>(dcat) + m * dcat_stride + cg * 8) = v;
        }
    }
}

}  // namespace

extern "C" int vn_heads_dgrad_f32(const float *d_rows, int64_t d_rows_stride, const float *w, void *d_cat, vnDtype d_cat_dtype,
                                  int64_t d_cat_stride, int64_t M, vnStream stream) {
    VN_CHECK_ARG(d_rows && w && d_cat && M > 0 && d_rows_stride >= 16 && (d_rows_stride & 3) == 0 && d_cat_stride >= 768 &&
                 (d_cat_stride & 7) == 0 && (d_cat_dtype == VN_F32 || d_cat_dtype == VN_BF16));
    if ((reinterpret_cast<uintptr_t>(d_rows) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(d_cat) & 15))
        return VN_EUNSUPPORTED;
    int64_t blocks = (M + 1) / 2;
    if (blocks > 4096) blocks = 4096;
    if (d_cat_dtype == VN_F32)
        k_heads_dgrad_f32<true><<<(unsigned)blocks, 192, 0, vn_stream(stream)>>>(d_rows, d_rows_stride, w, d_cat, d_cat_stride, M);
    else
        k_heads_dgrad_f32<false><<<(unsigned)blocks, 192, 0, vn_stream(stream)>>>(d_rows, d_rows_stride, w, d_cat, d_cat_stride, M);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// prob (B,2,S), reg (B,14,S) fp32 = the two heads over the (B*S, 768) bf16 concat rows; w = the packed [16][768] bf16 weight
// (rows 0-1: the score head, 2-15: the regression head; vn_pack_weight mode 0), bias [16] fp32 or NULL; sigmoid on the scores
extern "C" int vn_heads_fwd(const void *cat_rows, int64_t cat_stride, const void *w_packed, const float *bias, int32_t B, int64_t S,
                            float *prob, float *reg, vnStream stream) {
    VN_CHECK_ARG(cat_rows && w_packed && prob && reg && B > 0 && S > 0 && cat_stride >= 768 && (cat_stride & 7) == 0);
    if ((reinterpret_cast<uintptr_t>(cat_rows) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15) ||
        (reinterpret_cast<uintptr_t>(prob) & 15) || (reinterpret_cast<uintptr_t>(reg) & 15))
        return VN_EUNSUPPORTED;
    const int64_t groups = ((int64_t)B * S + 15) >> 4;
    int64_t blocks = (groups + 3) / 4;
    // (every wave loads the 24 KB of weights first: few, long-lived waves — two workgroups per CU)
    const int cap = 512;
    if (blocks > cap) blocks = cap;
    k_heads_fwd<<<(unsigned)blocks, 256, 0, vn_stream(stream)>>>(static_cast<const bf16_t *>(cat_rows), cat_stride,
                                                                 static_cast<const bf16_t *>(w_packed), bias, B, S, prob, reg);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// d_cat (M, 768) bf16 rows = d_rows (M, 16) bf16 . W;  w_packed_dgrad = the packed [768][16] bf16 weight (vn_pack_weight mode 1)
extern "C" int vn_heads_dgrad(const void *d_rows, int64_t d_rows_stride, const void *w_packed_dgrad, void *d_cat,
                              int64_t d_cat_stride, int64_t M, vnStream stream) {
    VN_CHECK_ARG(d_rows && w_packed_dgrad && d_cat && M > 0 && d_rows_stride >= 16 && (d_rows_stride & 3) == 0 &&
                 d_cat_stride >= 768 && (d_cat_stride & 7) == 0);
    if ((reinterpret_cast<uintptr_t>(d_rows) & 7) || (reinterpret_cast<uintptr_t>(w_packed_dgrad) & 7) ||
        (reinterpret_cast<uintptr_t>(d_cat) & 15))
        return VN_EUNSUPPORTED;
    const int64_t groups = (M + 15) >> 4;
    int64_t blocks = (groups + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    k_heads_dgrad<<<(unsigned)blocks, 256, 0, vn_stream(stream)>>>(static_cast<const bf16_t *>(d_rows), d_rows_stride,
                                                                   static_cast<const bf16_t *>(w_packed_dgrad),
                                                                   static_cast<bf16_t *>(d_cat), d_cat_stride, M);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
