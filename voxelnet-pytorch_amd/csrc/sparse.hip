// Sparsity support for the first middle layer (ConvMD(3,128,64,3,(2,1,1),(1,1,1)), model.py:207).
// Its input is the scattered voxel grid (model.py:102-106): ~6k occupied of 1.4 M sites per sample, so
//   * its forward differs from the bias only at output sites with an occupied voxel in their 3x3x3
//     receptive field ("active" sites, <= 18 per voxel),
//   * its weight gradient and the only part of its data gradient anybody reads (the rows the scatter's
//     backward gathers) are sums over the occupied voxels.
// This file builds the active-site list: mark (one thread per voxel x tap) + ORDERED compaction
// (block counts -> single-block exclusive scan -> per-block write), so the list order — and with it every
// downstream fp32 summation order — is deterministic.  The row-list modes of the two MFMA kernels
// (conv.hip, wgrad.hip) consume it.  HBM-bound, tiny: 1.4 MB of flags per batch item pair.
#include "common.h"

namespace {

constexpr int AS_THREADS = 256, AS_ITEMS = 8, AS_TILE = AS_THREADS * AS_ITEMS;

struct ASGeom {
    int32_t B, Do, Ho, Wo;
    int32_t kD, kH, kW, sD, sH, sW, pD, pH, pW;
};

__global__ void __launch_bounds__(256) k_mark(const int64_t *__restrict__ coord, int64_t K, ASGeom g,
                                              uint8_t *__restrict__ flags) {
    VN_PRIO_MAIN();
    const int taps = g.kD * g.kH * g.kW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * taps) return;
    const int64_t v = i / taps;
    const int t = (int)(i - v * taps);
    const int td = t / (g.kH * g.kW), th = (t / g.kW) % g.kH, tw = t % g.kW;
    const int64_t *c = coord + v * 4;
    const int b = (int)c[0];
    const int nd = (int)c[1] + g.pD - td, nh = (int)c[2] + g.pH - th, nw = (int)c[3] + g.pW - tw;
    if (nd < 0 || nh < 0 || nw < 0 || nd % g.sD || nh % g.sH || nw % g.sW) return;
    const int od = nd / g.sD, oh = nh / g.sH, ow = nw / g.sW;
    if (b < 0 || b >= g.B || od >= g.Do || oh >= g.Ho || ow >= g.Wo) return;
    flags[(((int64_t)b * g.Do + od) * g.Ho + oh) * g.Wo + ow] = 1;
}

__device__ int block_excl_scan_i32(int v, int *total) {
    __shared__ int wsum[AS_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < AS_THREADS / 64; ++w) {
        const int s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(AS_THREADS) k_as_count(const uint8_t *__restrict__ flags, int64_t n,
                                                         int32_t *__restrict__ blk) {
    VN_PRIO_MAIN();
    const int64_t base = (int64_t)blockIdx.x * AS_TILE + (int64_t)threadIdx.x * AS_ITEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < AS_ITEMS; ++j)
        if (base + j < n) s += flags[base + j] ? 1 : 0;
    int tot;
    block_excl_scan_i32(s, &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(AS_THREADS) k_as_scan(int32_t *__restrict__ blk, int64_t nb, int64_t cap,
                                                        int32_t *__restrict__ count) {
    VN_PRIO_MAIN();
    int carry = 0;
    for (int64_t base = 0; base < nb; base += AS_THREADS) {
        const int64_t i = base + threadIdx.x;
        const int v = i < nb ? blk[i] : 0;
        int tot;
        const int ex = block_excl_scan_i32(v, &tot);
        if (i < nb) blk[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) count[0] = carry < cap ? carry : (int32_t)cap;
}

__global__ void __launch_bounds__(AS_THREADS) k_as_write(const uint8_t *__restrict__ flags, int64_t n,
                                                         const int32_t *__restrict__ blk, ASGeom g,
                                                         int64_t *__restrict__ list, int64_t cap) {
    VN_PRIO_MAIN();
    const int64_t base = (int64_t)blockIdx.x * AS_TILE + (int64_t)threadIdx.x * AS_ITEMS;
    uint8_t f[AS_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < AS_ITEMS; ++j) {
        f[j] = base + j < n ? flags[base + j] : 0;
        s += f[j] ? 1 : 0;
    }
    int tot;
    int pos = block_excl_scan_i32(s, &tot) + blk[blockIdx.x];
#pragma unroll
    for (int j = 0; j < AS_ITEMS; ++j) {
        if (!f[j]) continue;
        if (pos < cap) {
            int64_t site = base + j;
            const int64_t ow = site % g.Wo; site /= g.Wo;
            const int64_t oh = site % g.Ho; site /= g.Ho;
            const int64_t od = site % g.Do;
            const int64_t b = site / g.Do;
            int64_t *o = list + (int64_t)pos * 4;
            o[0] = b; o[1] = od; o[2] = oh; o[3] = ow;
        }
        ++pos;
    }
}

__global__ void __launch_bounds__(256) k_fill_rows(void *__restrict__ y, int f32, int64_t M, int C, int64_t stride,
                                                   const float *__restrict__ bias) {
    VN_PRIO_MAIN();
    const int groups = C >> 2;
    const int64_t total = M * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / groups;
        const int c = (int)(i - m * groups) << 2;
        const float4 b = *reinterpret_cast<const float4 *>(bias + c);
        if (f32) {
            *reinterpret_cast<float4 *>(static_cast<float *>(y) + m * stride + c) = b;
        } else {
            bf16x4_t o;
            o[0] = (bf16_t)b.x; o[1] = (bf16_t)b.y; o[2] = (bf16_t)b.z; o[3] = (bf16_t)b.w;
            *reinterpret_cast<bf16x4_t *>(static_cast<bf16_t *>(y) + m * stride + c) = o;
        }
    }
}

// ---- "rulebook" evaluation of the first Conv3d ---------------------------------------------------------------------
// out[site] = bias + sum over the taps t whose source cell holds an occupied voxel v of  W[t] . x[v].
// Step 1 (a plain dense GEMM, vn_conv_gather_gemm on the (K,128) voxel rows against the [taps*Cout][Cin] packed
// weights): P[v][t][:] = W[t] . x[v] for every voxel and tap — 2*K*27*128*64 FLOP instead of the
// dense-equivalent 2*sites*27*128*64 over the ~10x more numerous active sites.
// Step 2 (here): every active site looks its <= 27 source cells up in an int32 voxel-index grid and adds the
// matching P rows in tap order (deterministic), writes y and the per-workgroup BatchNorm partial sums.
__global__ void __launch_bounds__(256) k_index_scatter(const int64_t *__restrict__ coord, int64_t K, int B, int D, int H,
                                                       int W, int32_t *__restrict__ grid) {
    VN_PRIO_MAIN();
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= K) return;
    const int64_t *c = coord + v * 4;
    const int64_t b = c[0], z = c[1], y = c[2], x = c[3];
    if (b < 0 || b >= B || z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;
    grid[((b * D + z) * H + y) * W + x] = (int32_t)v;
}

constexpr int RB_ROWS = 64;   // list rows per block of k_rulebook_combine
// A workgroup takes the blocks blockIdx.x, blockIdx.x + gridDim.x, ... and leaves ONE row of the statistics slab: the
// finalize that follows on the dependency chain sums at most RB_MAX_BLOCKS rows instead of cap / 64 (44,000 rows = 122 us
// at the dense configuration's 160 k voxels, 6,750 rows = 13 us at the car configuration's 24 k).  2048 workgroups x 4
// waves fill every wave slot of the chip, which is all the latency-bound per-site chain can use.
constexpr int RB_MAX_BLOCKS = 2048;

template <bool OUT_F32>
__global__ void __launch_bounds__(256) k_rulebook_combine(const float *__restrict__ P, const int32_t *__restrict__ grid,
                                                          const int64_t *__restrict__ list,
                                                          const int32_t *__restrict__ count, int64_t cap, ASGeom g, int Di,
                                                          int Hi, int Wi, int C, const float *__restrict__ bias,
                                                          void *__restrict__ y, float *__restrict__ slab) {
    VN_PRIO_MAIN();
    // block = RB_ROWS list rows; wave w takes rows w, w+4, ...; lane = output channel (C == 64).  One slab row per
    // workgroup.  The per-site chain (coordinates -> 27 index lookups -> P rows) is latency-bound: every wave slot busy.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int taps = g.kD * g.kH * g.kW;
    const int64_t cnt = count[0];
    const int64_t n = cnt < cap ? cnt : cap;     // (the persistent loop is bounded by the count: never past the list's capacity)
    const float bv = bias ? bias[lane] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t blk = blockIdx.x; blk * RB_ROWS < n; blk += gridDim.x) {
        for (int r = wave; r < RB_ROWS; r += 4) {
            const int64_t m = blk * RB_ROWS + r;
            if (m >= n) break;
            const int64_t *rc = list + m * 4;
            const int b = (int)rc[0], od = (int)rc[1], oh = (int)rc[2], ow = (int)rc[3];
            // lane t < taps: voxel index of tap t's source cell (or -1)
            int idx = -1;
            if (lane < taps) {
                const int kd = lane / (g.kH * g.kW), kh = (lane / g.kW) % g.kH, kw = lane % g.kW;
                const int sd = od * g.sD + kd - g.pD, sh = oh * g.sH + kh - g.pH, sw = ow * g.sW + kw - g.pW;
                if ((unsigned)sd < (unsigned)Di && (unsigned)sh < (unsigned)Hi && (unsigned)sw < (unsigned)Wi)
                    idx = grid[(((int64_t)b * Di + sd) * Hi + sh) * Wi + sw];
            }
            uint64_t live = __builtin_amdgcn_ballot_w64(idx >= 0);
            float acc = 0.f;
            while (live) {                       // ascending tap order: a fixed summation order
                const int t = __builtin_ctzll(live);
                live &= live - 1;
                const int v = __builtin_amdgcn_readlane(idx, t);
                acc += P[((int64_t)v * taps + t) * C + lane];
            }
            s1 += acc;
            s2 += acc * acc;
            const int64_t o = ((((int64_t)b * g.Do + od) * g.Ho + oh) * g.Wo + ow) * C + lane;
            if (OUT_F32) static_cast<float *>(y)[o] = acc + bv;
            else static_cast<bf16_t *>(y)[o] = (bf16_t)(acc + bv);
        }
    }
    if (slab) {
        __shared__ float red[2][4][64];
        red[0][wave][lane] = s1;
        red[1][wave][lane] = s2;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6;
            slab[((int64_t)blockIdx.x * 2 + which) * C + lane] =
                red[which][0][lane] + red[which][1][lane] + red[which][2][lane] + red[which][3][lane];
        }
    }
}

struct ASPlan {
    int64_t sites, nb;
    size_t off_flags, off_blk, bytes;
};
ASPlan as_plan(const vnConv *g) {
    ASPlan p{};
    p.sites = (int64_t)g->B * g->Dr * g->Hr * g->Wr;
    p.nb = vn_ceil_div(p.sites, AS_TILE);
    size_t off = 0;
    p.off_flags = off; off += vn_align((size_t)p.sites);
    p.off_blk = off; off += vn_align(sizeof(int32_t) * (size_t)(p.nb + 1));
    p.bytes = off;
    return p;
}
bool as_geom_ok(const vnConv *g) {
    return g && g->B > 0 && g->Dr > 0 && g->Hr > 0 && g->Wr > 0 && g->kD >= 1 && g->kH >= 1 && g->kW >= 1 &&
           g->mulD >= 1 && g->mulH >= 1 && g->mulW >= 1 && g->tmulD == 1 && g->tmulH == 1 && g->tmulW == 1 &&
           g->divD == 1 && g->divH == 1 && g->divW == 1;
}

}  // namespace

extern "C" size_t vn_active_sites_workspace_bytes(const vnConv *geom) {
    return as_geom_ok(geom) ? as_plan(geom).bytes : 0;
}

extern "C" int vn_active_sites(const int64_t *coord, int64_t K, const vnConv *geom, void *workspace,
                               size_t workspace_bytes, int64_t *list, int64_t cap, int32_t *count, vnStream stream) {
    VN_CHECK_ARG(as_geom_ok(geom) && workspace && list && count && K >= 0 && cap >= 0 && cap < (1ll << 31));
    VN_CHECK_ARG(coord || K == 0);
    const ASPlan pl = as_plan(geom);
    if (workspace_bytes < pl.bytes) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    uint8_t *flags = static_cast<uint8_t *>(workspace) + pl.off_flags;
    int32_t *blk = reinterpret_cast<int32_t *>(static_cast<char *>(workspace) + pl.off_blk);
    const ASGeom g{geom->B, geom->Dr, geom->Hr, geom->Wr, geom->kD, geom->kH, geom->kW,
                   geom->mulD, geom->mulH, geom->mulW, geom->padD, geom->padH, geom->padW};
    VN_HIP(hipMemsetAsync(flags, 0, (size_t)pl.sites, st));
    if (K > 0) {
        const int64_t n = K * g.kD * g.kH * g.kW;
        k_mark<<<(unsigned)vn_ceil_div(n, 256), 256, 0, st>>>(coord, K, g, flags);
        VN_LAUNCH_STATUS();
    }
    k_as_count<<<(unsigned)pl.nb, AS_THREADS, 0, st>>>(flags, pl.sites, blk);
    VN_LAUNCH_STATUS();
    k_as_scan<<<1, AS_THREADS, 0, st>>>(blk, pl.nb, cap, count);
    VN_LAUNCH_STATUS();
    k_as_write<<<(unsigned)pl.nb, AS_THREADS, 0, st>>>(flags, pl.sites, blk, g, list, cap);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_fill_rows(void *y, vnDtype dtype, int64_t M, int32_t C, int64_t stride, const float *values,
                            vnStream stream) {
    VN_CHECK_ARG(M >= 0 && C > 0 && (C & 3) == 0 && (stride & 3) == 0 && stride >= C);
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(y && values);
    int64_t blocks = vn_ceil_div(M * (C >> 2), 256);
    if (blocks > 8192) blocks = 8192;
    k_fill_rows<<<(unsigned)blocks, 256, 0, vn_stream(stream)>>>(y, dtype == VN_F32, M, C, stride, values);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_voxel_index_grid(const int64_t *coord, int64_t K, int32_t B, int32_t D, int32_t H, int32_t W,
                                   int32_t *grid, vnStream stream) {
    VN_CHECK_ARG(grid && K >= 0 && B > 0 && D > 0 && H > 0 && W > 0 && (coord || K == 0) && K < (1ll << 31));
    hipStream_t st = vn_stream(stream);
    VN_HIP(hipMemsetAsync(grid, 0xFF, sizeof(int32_t) * (size_t)B * D * H * W, st));   // -1 everywhere
    if (K > 0) {
        k_index_scatter<<<(unsigned)vn_ceil_div(K, 256), 256, 0, st>>>(coord, K, B, D, H, W, grid);
        VN_LAUNCH_STATUS();
    }
    return VN_OK;
}

extern "C" int vn_rulebook_combine(const float *P, const int32_t *index_grid, const int64_t *list, int64_t cap,
                                   const int32_t *count, const vnConv *geom, const float *bias, void *y,
                                   vnDtype y_dtype, float *stats_slab, vnStream stream) {
    VN_CHECK_ARG(P && index_grid && list && count && y && as_geom_ok(geom) && cap >= 0);
    VN_CHECK_ARG(geom->Cr == 64 && geom->kD * geom->kH * geom->kW <= 64);   // lane = channel / lane = tap
    VN_CHECK_ARG(y_dtype == VN_F32 || y_dtype == VN_BF16);
    if (cap == 0) return VN_OK;
    const ASGeom g{geom->B, geom->Dr, geom->Hr, geom->Wr, geom->kD, geom->kH, geom->kW,
                   geom->mulD, geom->mulH, geom->mulW, geom->padD, geom->padH, geom->padW};
    const unsigned blocks = (unsigned)vn_rulebook_slab_rows(cap);
    if (y_dtype == VN_F32)
        k_rulebook_combine<true><<<blocks, 256, 0, vn_stream(stream)>>>(P, index_grid, list, count, cap, g, geom->Ds, geom->Hs,
                                                                        geom->Ws, geom->Cr, bias, y, stats_slab);
    else
        k_rulebook_combine<false><<<blocks, 256, 0, vn_stream(stream)>>>(P, index_grid, list, count, cap, g, geom->Ds, geom->Hs,
                                                                         geom->Ws, geom->Cr, bias, y, stats_slab);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int64_t vn_rulebook_slab_rows(int64_t cap) {
    const int64_t b = cap > 0 ? vn_ceil_div(cap, RB_ROWS) : 0;
    return b > RB_MAX_BLOCKS ? RB_MAX_BLOCKS : b;
}
