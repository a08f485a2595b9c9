// Gather-GEMM convolution on the gfx950 matrix cores — replaces the ATen conv /
// conv_transpose calls behind model.py:111-199 (ConvMD, DeConv2d), forward and
// data-gradient.
//
//   out[m, n] = bias[n] + sum_{tap, k} src[site(m, tap), k] * w[tap][n][k]
//
// Design (CDNA4-first):
//   * One workgroup = 4 waves = a (64*WM) x (64*WN) output tile; every wave owns a
//     64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators (64 VGPRs).
//   * K loop = taps x (Cs/64).  Each step stages a [rows][64] bf16 slab of the
//     gathered source rows and of the packed weights in LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds): every lane supplies its own source address,
//     so the im2col gather costs no registers and no ds_write.  Invalid taps
//     (padding, out-of-range, residue mismatch) point past the buffer
//     descriptor's num_records and the hardware returns zeros.
//   * LDS rows are 128 B; the 16-B chunk c of row r sits at slot c ^ ((r>>1)&7)
//     so the ds_read_b128 fragment reads are bank-conflict free.  LDS-DMA writes
//     linearly, so the swizzle is applied to the per-lane SOURCE address.
//   * Two LDS stages: the loads of step s+1 fly while step s is on the MFMAs.
//   * Weight rows are staged permuted so that a lane's four accumulator tiles
//     hold four CONSECUTIVE output channels: the epilogue stores 8 B (bf16) or
//     16 B (fp32) per lane, 128/256 B contiguous per row.
//   * Strided transposed gathers (ConvTranspose2d forward, data-gradient of a
//     strided conv) run as residue classes (blockIdx.y) of stride-1 gathers.
//   * Optional fused BatchNorm reduction: per-channel sum / sum of squares of
//     the fp32 accumulators (before the bias is added), one partial row per
//     workgroup in a slab (no atomics), reduced in double by vn_bn_finalize.
#include "common.h"

namespace {

constexpr int GG_MAX_TAPS = 32;
constexpr int GG_MAX_CLASSES = 16;
constexpr uint32_t GG_OOB = 0xFFFFF000u;   // voffset of an invalid gather row
constexpr uint32_t GG_MAX_WINDOW = 0xFFFFE000u;

struct GGClass {
    int32_t tap_begin, ntaps;
    int32_t nD, nH, nW;            // taps per axis (taps are enumerated D-major: t = (id*nH + ih)*nW + iw)
    int32_t qD, qH, qW;            // row grid of this class
    int32_t ooffD, ooffH, ooffW;   // out coord = q * omul + ooff
    int32_t offD[4], offH[4], offW[4];   // src coord = q * mul + off[axis index]
};
struct GGTap {
    int32_t id, ih, iw, widx;
};
struct GGParams {
    const char *src;
    const char *w;
    const float *bias;
    char *out;
    float *stats;               // [tiles_m][2][N] per-workgroup partial sums, or NULL
    // data-gradient launches (k_conv_patch2d): stats = the BatchNorm-BACKWARD sums of the layer below instead — its conv
    // output bn_y (same site layout as out), its forward statistics bn_stats [4][N]: sum dz, sum dz*xhat with
    // dz = relu-masked out value as stored (what vn_bn_bwd_reduce_slab would read back)
    const void *bn_y;
    const float *bn_stats;
    int32_t bn_y_f32;
    int64_t sB, sD, sH, sW;   // elements
    int64_t oB, oD, oH, oW;
    int32_t B, Ds, Hs, Ws, Do, Ho, Wo;
    int32_t mulD, mulH, mulW, omulD, omulH, omulW;
    int32_t Cs, src_wrap, N, out_f32, accumulate, nclasses, src_row_elems;
    int32_t esz;                // operand element size: 2 (bf16) or 4 (fp32, exact v_mfma_f32_16x16x4_f32 path)
    int32_t x3;                 // fp32 storage, products as three bf16 MFMAs (VN_F32X3): the F32 kernels' alternative inner loop;
                                // 2 = the packed weights hold hi / lo bf16 granules (Cs % 32 == 0), 1 = fp32 weights split in registers,
                                // 3 = as 2 and the SOURCE rows are stored split as well (VN_F32X3S: no split pass at all)
    uint32_t w_bytes;
    int64_t src_batch_extent;   // elements spanned by one batch item (for num_records)
    // row-list mode (sparse first Conv3d): rows are an explicit list of (b,d,h,w) coordinates
    const int64_t *row_list;    // [row_cap][4] int64, or NULL (dense rows)
    const int32_t *row_count;   // device: number of valid list rows, or NULL (= row_cap)
    int32_t row_cap;
    int32_t out_linear;         // list mode: output row offset = row index * oW (else from the coordinates)
    int32_t ldivD, ldivH, ldivW;             // list mode: row coordinate = div*q + r
    int32_t slotcD[4], slotcH[4], slotcW[4]; // list mode: residue r a row must have for offset slot j (-1: any)
    GGClass cls[GG_MAX_CLASSES];
    GGTap taps[GG_MAX_TAPS];
};

typedef __attribute__((address_space(3))) void lds_void_t;
#ifdef VN_P2D_TRACE
// -DVN_P2D_TRACE builds only (tools/trace_patch2d.py): shader-clock stamps of wave 0 of workgroup (1, 0) of every
// k_conv_patch2d launch, per tap step: [0] before the wait, [1] after it, [2] after the barrier, [3] after the DMA issue,
// [4] after the fragment reads + MFMAs (a wait for the accumulators)
__device__ long long g_p2d_trace[64 * 8];
#define P2D_STAMP(s, k) do { if (tr_on) g_p2d_trace[((s) & 63) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P2D_STAMP(s, k) do {} while (0)
#endif
#ifndef GG_PIN_SCHEDULE
#define GG_PIN_SCHEDULE 1
#endif
#if GG_PIN_SCHEDULE
#define GG_PIN() __builtin_amdgcn_sched_barrier(0)   // keep a DMA piece between its MFMA groups
#else
#define GG_PIN()
#endif

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char *lds_wave_base, uint32_t voffset,
                                          uint32_t soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// fp32x3: a lane's eight weight values (channels 8 fq .. 8 fq + 7 of a 32-channel chunk: granules 2 fq and 2 fq + 1) as hi /
// lo bf16 — the two granules vn_pack_weight(VN_F32X3) wrote (presplit) or split here from the two fp32 granules
__device__ __forceinline__ void x3_weights(bool presplit, const char *g0, const char *g1, bf16x8_t &hi, bf16x8_t &lo) {
    if (presplit) {
        hi = *reinterpret_cast<const bf16x8_t *>(g0);
        lo = *reinterpret_cast<const bf16x8_t *>(g1);
    } else {
        vn_split8(*reinterpret_cast<const f32x4_t *>(g0), *reinterpret_cast<const f32x4_t *>(g1), hi, lo);
    }
}

// fp32x3, patch kernels: the staged fp32 patch (rows of 128 B = 32 channels, 16-B granules XOR-swizzled by (row >> 1) & 7)
// rewritten IN PLACE as hi / lo bf16 granules, once per (plane, chunk) by the whole workgroup — the granule pair (2 f, 2 f + 1)
// = the eight fp32 values of channels 8 f .. 8 f + 7 becomes their eight hi parts (granule 2 f) and eight lo parts (granule
// 2 f + 1): the "split fp32" format (VN_F32X3S) that x3_weights() reads of a pre-split weight row and that a pre-split
// SOURCE (p.x3 == 3: the BatchNorm passes wrote it that way, round 5) arrives in — then this pass is skipped.  The nine taps
// (x WN waves) read every A fragment ready made instead of splitting it again: 24 VALU operations per fragment and use were
// what bounded these kernels (round 4).
template <int NT>
__device__ __forceinline__ void x3_split_patch(char *patch, int rows) {
    for (int item = threadIdx.x; item < rows * 4; item += NT) {
        const int q = item >> 2, f = item & 3, sw = (q >> 1) & 7;
        char *g0 = patch + q * 128 + (((2 * f) ^ sw) << 4), *g1 = patch + q * 128 + (((2 * f + 1) ^ sw) << 4);
        bf16x8_t hi, lo;
        vn_split8(*reinterpret_cast<const f32x4_t *>(g0), *reinterpret_cast<const f32x4_t *>(g1), hi, lo);
        *reinterpret_cast<bf16x8_t *>(g0) = hi;
        *reinterpret_cast<bf16x8_t *>(g1) = lo;
    }
}

// Store one workgroup's accumulators (+bias, optional accumulate) through the per-row offset table otab (LDS, -1 =
// row not stored) and, if asked, its per-channel sum / sum of squares into stats slab row `tile`.
template <int WM, int WN, int SM, bool BNBWD = false>
__device__ __forceinline__ void gg_store(const GGParams &p, f32x4_t (&acc)[SM][4], char *smem, const int32_t *otab, int64_t tile,
                                         int n0, int wm, int wn, int lane) {
    constexpr int BN = 64 * WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t tile_m = tile;
    const int ncol = n0 + wn * 64 + fr * 4;   // this lane's 4 consecutive output channels
    const bool col_ok = ncol < p.N;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && col_ok) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.bias + ncol);
        bias4[0] = bv.x; bias4[1] = bv.y; bias4[2] = bv.z; bias4[3] = bv.w;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float bmean[4] = {0.f, 0.f, 0.f, 0.f}, binv[4] = {0.f, 0.f, 0.f, 0.f}, bS[4] = {0.f, 0.f, 0.f, 0.f}, bbe[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BNBWD) {
        if (col_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bmean[j] = p.bn_stats[ncol + j]; binv[j] = p.bn_stats[p.N + ncol + j];
                bS[j] = p.bn_stats[2 * p.N + ncol + j]; bbe[j] = p.bn_stats[3 * p.N + ncol + j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int4 offs = *reinterpret_cast<const int4 *>(otab + wm * (16 * SM) + i * 16 + fq * 4);
        const int32_t o4[4] = {offs.x, offs.y, offs.z, offs.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (o4[e] < 0 || !col_ok) continue;
            float v[4];
            if constexpr (BNBWD) {
                // (no bias, no accumulate on this path: checked on the host)  dz of the rounded value, as the separate pass reads it
                float yv[4];
                if (p.bn_y_f32) {
                    const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.bn_y) + (int64_t)o4[e] + ncol);
                    yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w;
                } else {
                    const bf16x4_t t = *reinterpret_cast<const bf16x4_t *>(reinterpret_cast<const bf16_t *>(p.bn_y) + (int64_t)o4[e] + ncol);
#pragma unroll
                    for (int j = 0; j < 4; ++j) yv[j] = (float)t[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[i][j][e];
                    v[j] = a;
                    const float st = p.out_f32 ? a : (float)(bf16_t)a;
                    const float d0 = yv[j] - bmean[j];
                    const float z = fmaf(bS[j], d0, bbe[j]);
                    const float dz = z > 0.f ? st : 0.f;
                    s1[j] += dz;
                    s2[j] += dz * (d0 * binv[j]);
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = acc[i][j][e];
                s1[j] += a;
                s2[j] += a * a;
                v[j] = a + bias4[j];
            }
            }
            if (p.out_f32) {
                float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.out) + (int64_t)o4[e] + ncol);
                if (p.accumulate) {
                    const float4 old = *dst;
                    v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
                }
                *dst = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                bf16x4_t *dst = reinterpret_cast<bf16x4_t *>(reinterpret_cast<bf16_t *>(p.out) + (int64_t)o4[e] + ncol);
                if (p.accumulate) {
                    const bf16x4_t old = *dst;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)old[j];
                }
                bf16x4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
                *dst = o;
            }
        }
    }
    if (p.stats) {
        // per-workgroup partial sums -> slab[tile_m][2][N] (plain stores, no atomics: thousands of waves adding
        // into the same 2N addresses serialise at the memory side; vn_bn_finalize reduces the slab in double)
        __syncthreads();                                  // otab reads done; reuse LDS
        float *red = reinterpret_cast<float *>(smem);     // [2][BN] per M-wave group
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] += __shfl_xor(s1[j], 16, 64);
            s1[j] += __shfl_xor(s1[j], 32, 64);
            s2[j] += __shfl_xor(s2[j], 16, 64);
            s2[j] += __shfl_xor(s2[j], 32, 64);
        }
        if (fq == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[(wm * 2 + 0) * BN + wn * 64 + fr * 4 + j] = s1[j];
                red[(wm * 2 + 1) * BN + wn * 64 + fr * 4 + j] = s2[j];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * BN; i += 64 * WM * WN) {
            const int which = i / BN, c = i - which * BN;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) v += red[(w * 2 + which) * BN + c];
            if (n0 + c < p.N) p.stats[((int64_t)tile_m * 2 + which) * p.N + n0 + c] = v;
        }
    }
}

// WM x WN waves, each owning a (16*SM) x 64 sub-tile; NS LDS stages (the loads run NS-1 steps ahead of the MFMAs:
// the small late layers — 4400 rows x 2304 K — are latency-bound, not bandwidth-bound, and want a deep pipeline)
template <int WM, int WN, int SM, int NS, bool F32>
__global__ void __launch_bounds__(256, 2) k_gather_gemm(const GGParams p) {
    VN_PRIO_MAIN();
    static_assert(WM * WN == 4 && NS >= 2 && SM >= 1 && SM <= 8, "4 waves");
    constexpr int ESZ = F32 ? 4 : 2;          // a K step is always 128 B of every row: 64 bf16 or 32 fp32
    constexpr int BM = 16 * SM * WM, BN = 64 * WN;
    static_assert(BM % 32 == 0, "a DMA instruction of the 4 waves covers 32 rows");
    constexpr int RA = BM / 32, RB = BN / 32;     // LDS-DMA instructions per wave per step
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int AHEAD = (NS - 2) * (RA + RB);   // DMA instructions of later stages that may still be in flight
    static_assert(AHEAD <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- tile of this workgroup (XCD-aware: neighbouring tiles share an L2) ----
    const GGClass &cl = p.cls[blockIdx.y];
    const int ntn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / ntn, tile_n = bid - tile_m * ntn;
    const int qD = cl.qD, qH = cl.qH, qW = cl.qW;
    const bool list = p.row_list != nullptr;
    const int64_t Mq = list ? (p.row_count ? (int64_t)p.row_count[0] : (int64_t)p.row_cap) : (int64_t)p.B * qD * qH * qW;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    if (m0 >= Mq) {   // block-uniform
        if (p.stats)    // capacity launch: the slab rows of empty tiles must read as zero
            for (int i = threadIdx.x; i < 2 * BN; i += 256) {
                const int which = i / BN, c = i - which * BN;
                if (n0 + c < p.N) p.stats[(((int64_t)blockIdx.y * (gridDim.x / ntn) + tile_m) * 2 + which) * p.N + n0 + c] = 0.f;
            }
        return;
    }
    const int rows_per_b = qD * qH * qW;
    const int b0 = list ? 0 : (int)(m0 / rows_per_b);

    // ---- per-lane loader state ----
    // A rows: instruction i of this wave covers LDS rows ((i*4+wave)*8 .. +7); lane -> row +(lane>>3), slot lane&7
    const int a_chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);   // source chunk for this lane's slot
    uint32_t a_row[RA];
    uint32_t a_bits[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int r = (i * 4 + wave) * 8 + (lane >> 3);
        const int64_t m = m0 + r;
        a_row[i] = GG_OOB;
        a_bits[i] = 0;
        if (m < Mq && list) {
            const int64_t *rc = p.row_list + m * 4;
            const int db = (int)rc[0], cd = (int)rc[1], ch = (int)rc[2], cw = (int)rc[3];
            const int qd = cd / p.ldivD, rd = cd - qd * p.ldivD;
            const int qh = ch / p.ldivH, rh = ch - qh * p.ldivH;
            const int qw = cw / p.ldivW, rw = cw - qw * p.ldivW;
            const int sd = qd * p.mulD, sh = qh * p.mulH, sw = qw * p.mulW;
            const int64_t e = (int64_t)db * p.sB + (int64_t)sd * p.sD + (int64_t)sh * p.sH + (int64_t)sw * p.sW;
            a_row[i] = (uint32_t)(e * ESZ);
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bits |= (((p.slotcD[j] < 0 || p.slotcD[j] == rd) && (uint32_t)(sd + cl.offD[j]) < (uint32_t)p.Ds) ? 1u : 0u) << j;
                bits |= (((p.slotcH[j] < 0 || p.slotcH[j] == rh) && (uint32_t)(sh + cl.offH[j]) < (uint32_t)p.Hs) ? 1u : 0u) << (4 + j);
                bits |= (((p.slotcW[j] < 0 || p.slotcW[j] == rw) && (uint32_t)(sw + cl.offW[j]) < (uint32_t)p.Ws) ? 1u : 0u) << (8 + j);
            }
            a_bits[i] = bits;
        } else if (m < Mq) {
            int t = (int)(m - (int64_t)b0 * rows_per_b);   // row index relative to batch b0 (may span batches)
            const int db = t / rows_per_b;
            t -= db * rows_per_b;
            const int qw = t % qW;
            t /= qW;
            const int qh = t % qH;
            const int qd = t / qH;
            const int sd = qd * p.mulD, sh = qh * p.mulH, sw = qw * p.mulW;
            const int64_t e = (int64_t)db * p.sB + (int64_t)sd * p.sD + (int64_t)sh * p.sH + (int64_t)sw * p.sW;
            a_row[i] = (uint32_t)(e * ESZ);
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bits |= ((uint32_t)(sd + cl.offD[j]) < (uint32_t)p.Ds ? 1u : 0u) << j;
                bits |= ((uint32_t)(sh + cl.offH[j]) < (uint32_t)p.Hs ? 1u : 0u) << (4 + j);
                bits |= ((uint32_t)(sw + cl.offW[j]) < (uint32_t)p.Ws ? 1u : 0u) << (8 + j);
            }
            a_bits[i] = bits;
        }
    }
    // B rows: LDS row rho of the tile holds weight row n0 + (rho/64)*64 + (rho%16)*4 + (rho%64)/16
    uint32_t b_row[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int rho = (i * 4 + wave) * 8 + (lane >> 3);
        const int rl = rho & 63;
        const int n = n0 + (rho & ~63) + (rl & 15) * 4 + (rl >> 4);
        b_row[i] = n < p.N ? (uint32_t)((int64_t)n * p.Cs * ESZ) : GG_OOB;   // chunk = a_chunk for every i
    }

    // buffer descriptors (wave-uniform by construction: kernel args and blockIdx only)
    const char *src_base = p.src + (int64_t)b0 * p.sB * ESZ;
    int64_t src_bytes = ((int64_t)(p.B - b0 - 1) * p.sB + p.src_batch_extent) * ESZ;
    if (src_bytes > (int64_t)GG_MAX_WINDOW) src_bytes = GG_MAX_WINDOW;
    const __amdgpu_buffer_rsrc_t rs_a = vn_uniform_rsrc(src_base, (uint32_t)src_bytes);
    const __amdgpu_buffer_rsrc_t rs_b = vn_uniform_rsrc(p.w, p.w_bytes);

    constexpr int BKE = 128 / ESZ;            // elements per K step
    constexpr int EPC = 16 / ESZ;             // elements per 16-B chunk
    const int nk = (p.Cs + BKE - 1) / BKE;    // the last step may be partial: chunks past Cs read zeros

    // taps that are invalid for EVERY row of this tile (whole padding planes: the tile lies in the first/last
    // depth plane or image line) are skipped block-uniformly
    uint32_t tapmask = 0;
    {
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) mine |= a_bits[i];
        uint32_t wavebits = 0;                // OR over the wave: one ballot per validity bit
#pragma unroll
        for (int b = 0; b < 12; ++b) wavebits |= (__builtin_amdgcn_ballot_w64((mine >> b) & 1u) != 0 ? 1u : 0u) << b;
        // the last 16 bytes of the last LDS stage are not written before the first loop barrier has been passed
        uint32_t *words = reinterpret_cast<uint32_t *>(smem + NS * STAGE - 16);
        if (lane == 0) words[wave] = wavebits;
        __syncthreads();
        const uint32_t blockbits = __builtin_amdgcn_readfirstlane(words[0] | words[1] | words[2] | words[3]);
        int t = 0;
        for (int id = 0; id < cl.nD; ++id)
            for (int ih = 0; ih < cl.nH; ++ih)
                for (int iw = 0; iw < cl.nW; ++iw, ++t) {
                    const uint32_t tb = (1u << id) | (16u << ih) | (256u << iw);
                    if ((blockbits & tb) == tb) tapmask |= 1u << t;
                }
    }
    const int nsteps = __builtin_popcount(tapmask) * nk;

    // One stage = RA + RB LDS-DMA pieces per wave (1 KiB each: 8 rows x 128 B).  The TA path moves ~40 B/clk/CU at
    // best, i.e. a piece occupies it for ~25-100 clk: the pieces of stage s+NS-1 are therefore issued one by one
    // BETWEEN the MFMA groups of step s (piece()), not in a burst in front of them.
    struct StageCtx {
        uint32_t tapbits, a_off, b_koff, b_soff;
        bool k_ok;
        char *la, *lb;
    };
    auto prep = [&](int tap, int kc, int buf, bool live) {
        const GGTap tp = p.taps[cl.tap_begin + tap];
        StageCtx c;
        c.tapbits = (1u << tp.id) | (16u << tp.ih) | (256u << tp.iw);
        const int64_t de = (int64_t)cl.offD[tp.id] * p.sD + (int64_t)cl.offH[tp.ih] * p.sH +
                           (int64_t)cl.offW[tp.iw] * p.sW;
        // this lane's K position inside the step
        const int k_lane = kc * BKE + a_chunk * EPC;
        c.k_ok = live && k_lane < p.Cs;
        const int k_src = k_lane;
        c.a_off = (uint32_t)(int32_t)(de * ESZ) + (uint32_t)k_src * (uint32_t)ESZ;
        c.b_koff = (uint32_t)k_lane * (uint32_t)ESZ;
        c.b_soff = __builtin_amdgcn_readfirstlane((uint32_t)((int64_t)tp.widx * p.N * p.Cs * ESZ));
        c.la = smem + buf * STAGE + wave * 1024;
        c.lb = smem + buf * STAGE + A_BYTES + wave * 1024;
        return c;
    };
    auto piece = [&](const StageCtx &c, int idx) {   // idx is a compile-time constant after unrolling
        if (idx < RA) {
            const bool ok = c.k_ok && (a_bits[idx] & c.tapbits) == c.tapbits;
            lds_dma16(rs_a, c.la + idx * 4096, ok ? a_row[idx] + c.a_off : GG_OOB, 0);
        } else {
            const int i = idx - RA;
            lds_dma16(rs_b, c.lb + i * 4096, (c.k_ok && b_row[i] != GG_OOB) ? b_row[i] + c.b_koff : GG_OOB, c.b_soff);
        }
    };
    // iterator over the (valid tap, K chunk) steps, one step ahead of the compute
    uint32_t it_mask = tapmask;
    int it_tap = tapmask ? __builtin_ctz(tapmask) : 0, it_kc = 0;
    auto advance = [&]() {
        if (++it_kc == nk) {
            it_kc = 0;
            it_mask &= it_mask - 1;
            it_tap = it_mask ? __builtin_ctz(it_mask) : 0;
        }
    };

    f32x4_t acc[SM][4];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row (lane&15), chunk (ks*4 + lane>>4) ^ ((lane&15)>>1)
    const int fr = lane & 15, fq = lane >> 4;
    const int frag_off0 = fr * 128 + (((0 + fq) ^ (fr >> 1)) << 4);
    const int frag_off1 = fr * 128 + (((4 + fq) ^ (fr >> 1)) << 4);
    // fp32x3: the lane's eight k values are channels 8 fq .. 8 fq + 7 of the 32-channel chunk — granules 2 fq, 2 fq + 1
    const int x3_off0 = fr * 128 + (((2 * fq) ^ (fr >> 1)) << 4);
    const int x3_off1 = fr * 128 + (((2 * fq + 1) ^ (fr >> 1)) << 4);

    constexpr int LPS = RA + RB;                // pieces per stage and wave
    // every stage slot is always issued (past the last step as out-of-range pieces: zeros, no memory traffic), so
    // exactly (NS-2)*LPS younger pieces are outstanding whenever stage s has to have landed
#pragma unroll
    for (int j = 0; j < NS - 1; ++j) {
        const StageCtx c = prep(it_tap, it_kc, j, j < nsteps);
#pragma unroll
        for (int q = 0; q < LPS; ++q) piece(c, q);
        if (j < nsteps) advance();
    }
    int buf = 0, nbuf = NS - 1;                 // stage consumed by this step / stage the next loads go to
    for (int s = 0; s < nsteps; ++s) {
        // RAW: stage s is read after ITS loads have landed in every wave (counted vmcnt, then the barrier).
        // WAR: right after the barrier the fastest wave re-stages the buffer step s-1 read — so every wave's LDS reads
        // of step s-1 must have RETURNED before it arrives: lgkmcnt(0).  (hipcc may sink the last MFMAs of a step, and
        // with them the wait for their operands, below an asm statement; without this wait the fragment reads of a slow
        // wave could still be in flight when the DMA overwrote them — one wrong tile in ~10 % of the steps once another
        // kernel shared the CUs, found by tools/debug_determinism.py.)
        // bare s_barrier: __syncthreads() carries a workgroup fence, which makes hipcc drain vmcnt to 0 (the LDS-DMA
        // loads of the stages still in flight) and so serialises the pipeline.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AHEAD) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool live = s + NS - 1 < nsteps;
        const StageCtx c = prep(it_tap, it_kc, nbuf, live);   // into the stage step s-1 has just finished with
        if (live) advance();
        const char *la = smem + buf * STAGE + wm * (16 * SM * 128);
        const char *lb = smem + buf * STAGE + A_BYTES + wn * (64 * 128);
        bool x3_done = false;
        if constexpr (F32) {
            if (p.x3) {      // both 16-B halves of the 32-channel chunk at once: the lane's eight k values, split hi / lo
                bf16x8_t ah[SM], al[SM], bh[4], bl[4];
#pragma unroll
                for (int i = 0; i < SM; ++i)      // (x3 == 3: the source rows are stored split — VN_F32X3S)
                    x3_weights(p.x3 == 3, la + i * 2048 + x3_off0, la + i * 2048 + x3_off1, ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) x3_weights(p.x3 >= 2, lb + j * 2048 + x3_off0, lb + j * 2048 + x3_off1, bh[j], bl[j]);
#pragma unroll
                for (int q = 0; q < LPS; ++q) piece(c, q);
#pragma unroll
                for (int i = 0; i < SM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = vn_mfma_x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
                x3_done = true;
            }
        }
        if (!x3_done) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? frag_off1 : frag_off0;
            if constexpr (F32) {
                // 16 B per lane = 4 consecutive k of one row; MFMA step t consumes element t of every lane's
                // vector (the k <-> (lane>>4, t) assignment is the same for A and B, which is all the sum needs)
                f32x4_t a[SM], b[4];
#pragma unroll
                for (int i = 0; i < SM; ++i) a[i] = *reinterpret_cast<const f32x4_t *>(la + i * 2048 + fo);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4_t *>(lb + j * 2048 + fo);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < SM; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < LPS; ++q)
                        if (q * 8 / LPS == ks * 4 + t) piece(c, q);
                }
            } else {
                bf16x8_t a[SM], b[4];
#pragma unroll
                for (int i = 0; i < SM; ++i) a[i] = *reinterpret_cast<const bf16x8_t *>(la + i * 2048 + fo);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t *>(lb + j * 2048 + fo);
#pragma unroll
                for (int i = 0; i < SM; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < LPS; ++q)
                        if (q * (2 * SM) / LPS == ks * SM + i) piece(c, q);
                    GG_PIN();
                }
            }
        }
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }

    // ---- epilogue ----
    // per-row output element offsets into LDS (stage memory is free now)
    __syncthreads();
    int32_t *otab = reinterpret_cast<int32_t *>(smem);
    for (int r = threadIdx.x; r < BM; r += 256) {
        const int64_t m = m0 + r;
        int32_t off = -1;
        if (m < Mq && list) {
            if (p.out_linear) {
                off = (int32_t)(m * p.oW);
            } else {
                const int64_t *rc = p.row_list + m * 4;
                off = (int32_t)(rc[0] * p.oB + rc[1] * p.oD + rc[2] * p.oH + rc[3] * p.oW);
            }
        } else if (m < Mq) {
            int t = (int)(m % rows_per_b);
            const int b = (int)(m / rows_per_b);
            const int qw = t % qW;
            t /= qW;
            const int qh = t % qH;
            const int qd = t / qH;
            const int od = qd * p.omulD + cl.ooffD, oh = qh * p.omulH + cl.ooffH, ow = qw * p.omulW + cl.ooffW;
            if (od < p.Do && oh < p.Ho && ow < p.Wo)
                off = (int32_t)((int64_t)b * p.oB + (int64_t)od * p.oD + (int64_t)oh * p.oH + (int64_t)ow * p.oW);
        }
        otab[r] = off;
    }
    __syncthreads();

    // slab row: residue classes (blockIdx.y) one after the other, gridDim.x / ntn rows each
    gg_store<WM, WN, SM>(p, acc, smem, otab, (int64_t)blockIdx.y * (gridDim.x / ntn) + tile_m, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 (x kD), stride-1-in-H/W convolutions with the im2col done INSIDE LDS ("patch" kernel).
// In k_gather_gemm every tap re-stages its A rows: 9 (27) x the source bytes through LDS-DMA, and the step time of
// that kernel is the DMA issue time (measured by an s_memtime trace: ~110-130 clk per 1-KiB piece and wave, not
// overlappable with the same wave's MFMAs).  Here a workgroup owns a TH x TW rectangle of output pixels of one
// (batch, depth) plane; per source plane and K chunk it stages the (TH+2) x (TW+2) halo patch ONCE and the nine
// (kh,kw) taps read their A fragments from it with shifted LDS row addresses.  Image borders are patch rows that
// were never in range: LDS-DMA wrote zeros there.  Only the 64-row weight tile is staged per tap.
// A pieces per tap step: (TH+2)(TW+2)/(8*9*4) per wave (~0.8) instead of BM/32 (5-8).
template <int WM, int WN, int SM, int TW, bool F32>
__global__ void __launch_bounds__(256, 2) k_conv_patch(const GGParams p) {
    VN_PRIO_MAIN();
    constexpr int ESZ = F32 ? 4 : 2;
    constexpr int BM = 16 * SM * WM, BN = 64 * WN, TH = BM / TW;
    static_assert(WM * WN == 4 && BM % TW == 0, "4 waves; whole patch lines");
    constexpr int PW = TW + 2, PH = TH + 2, PROWS = PH * PW;
    constexpr int PPIECES = (PROWS + 7) / 8;                 // 1-KiB pieces of the patch
    constexpr int PA = (PPIECES + 3) / 4;                    // per wave
    constexpr int PATCH_BYTES = PA * 4 * 1024;
    constexpr int RB = BN / 32;                              // weight pieces per wave and tap
    constexpr int B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *patch = smem;
    char *bst = smem + PATCH_BYTES;                          // two weight stages

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const GGClass &cl = p.cls[blockIdx.y];
    const int ntn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tile = bid / ntn, tile_n = bid - tile * ntn;
    const int n0 = tile_n * BN;
    const int tiles_x = (cl.qW + TW - 1) / TW, tiles_y = (cl.qH + TH - 1) / TH;
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int qd = t % cl.qD;
    const int b = t / cl.qD;
    if (b >= p.B) return;                                    // (a residue class with fewer planes than the largest)
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- loader state: patch rows of this lane (fixed over the planes and K chunks) and its weight rows
    const int a_chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    uint32_t a_row[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int q = (i * 4 + wave) * 8 + (lane >> 3);       // patch row
        const int qy = q / PW, qx = q - qy * PW;
        const int sy = y0 - 1 + qy, sx = x0 - 1 + qx;          // H/W: stride 1, offsets -1..1 (checked on the host)
        a_row[i] = (q < PROWS && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
                       ? (uint32_t)(((int64_t)sy * p.sH + (int64_t)sx * p.sW) * ESZ)
                       : GG_OOB;
    }
    uint32_t b_row[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int rho = (i * 4 + wave) * 8 + (lane >> 3);
        const int rl = rho & 63;
        const int n = n0 + (rho & ~63) + (rl & 15) * 4 + (rl >> 4);
        b_row[i] = n < p.N ? (uint32_t)((int64_t)n * p.Cs * ESZ) : GG_OOB;
    }
    const char *src_base = p.src + (int64_t)b * p.sB * ESZ;
    int64_t src_bytes = p.src_batch_extent * ESZ;
    if (src_bytes > (int64_t)GG_MAX_WINDOW) src_bytes = GG_MAX_WINDOW;
    const __amdgpu_buffer_rsrc_t rs_a = vn_uniform_rsrc(src_base, (uint32_t)src_bytes);
    const __amdgpu_buffer_rsrc_t rs_b = vn_uniform_rsrc(p.w, p.w_bytes);
    constexpr int BKE = 128 / ESZ, EPC = 16 / ESZ;
    const int nk = (p.Cs + BKE - 1) / BKE;
    const int k_lane0 = a_chunk * EPC;

    // ---- fragment geometry: MFMA row (lane & 15) of sub-tile i is output pixel (py, px) -> patch row of tap (0,0)
    const int fr = lane & 15, fq = lane >> 4;
    int q0[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int r = wm * (16 * SM) + i * 16 + fr;
        const int py = r / TW, px = r - py * TW;
        q0[i] = (py + 1) * PW + (px + 1);
    }
    const int bfrag0 = fr * 128 + (((0 + fq) ^ (fr >> 1)) << 4);
    const int bfrag1 = fr * 128 + (((4 + fq) ^ (fr >> 1)) << 4);
    const int x3_b0 = fr * 128 + (((2 * fq) ^ (fr >> 1)) << 4);        // fp32x3: granules 2 fq (hi) / 2 fq + 1 (lo)
    const int x3_b1 = fr * 128 + (((2 * fq + 1) ^ (fr >> 1)) << 4);

    f32x4_t acc[SM][4];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // H/W have one class and their taps in kernel order (div 1), so the weight index of tap (id, ih, iw) is that of
    // (id, 0, 0) plus ih*3 + iw: one scalar load per depth tap, none inside the tap loop
    const uint32_t tap_bytes = (uint32_t)p.N * (uint32_t)p.Cs * (uint32_t)ESZ;
    auto wbase_of = [&](int id) { return p.taps[cl.tap_begin + id * 9].widx; };
    auto stage_b = [&](int wbase, int tap, int kc, int buf) {
        const int k_lane = kc * BKE + k_lane0;
        const bool k_ok = k_lane < p.Cs;
        const uint32_t b_koff = (uint32_t)k_lane * (uint32_t)ESZ;
        const uint32_t b_soff = __builtin_amdgcn_readfirstlane((uint32_t)(wbase + tap) * tap_bytes);
        char *lb = bst + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < RB; ++i)
            lds_dma16(rs_b, lb + i * 4096, (k_ok && b_row[i] != GG_OOB) ? b_row[i] + b_koff : GG_OOB, b_soff);
    };
    auto stage_patch = [&](int sd, int kc) {
        const int k_lane = kc * BKE + k_lane0;
        const bool k_ok = k_lane < p.Cs;
        const int k_src = k_lane;
        const uint32_t off = (uint32_t)(((int64_t)sd * p.sD + k_src) * ESZ);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            lds_dma16(rs_a, patch + (i * 4 + wave) * 1024, (k_ok && a_row[i] != GG_OOB) ? a_row[i] + off : GG_OOB, 0);
    };

    int bbuf = 0;
    bool first = true;
    for (int id = 0; id < cl.nD; ++id) {
        const int sd = qd * p.mulD + cl.offD[id];
        if ((unsigned)sd >= (unsigned)p.Ds) continue;          // a padding plane: block-uniform skip
        const int wbase = wbase_of(id);
        for (int kc = 0; kc < nk; ++kc) {
            // everyone is done with the previous patch (and the weight stage the first tap's loads go to)
            __syncthreads();
            stage_patch(sd, kc);
            if (first) { stage_b(wbase, 0, kc, bbuf); first = false; }
            for (int tap = 0; tap < 9; ++tap) {
                const int ih = tap / 3, iw = tap - ih * 3;
                // (lgkmcnt(0): the previous tap's fragment reads of every wave have returned before the barrier lets the
                //  fastest wave re-stage that weight buffer — see k_gather_gemm)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // next weight tile: next tap, or the first tap of the next (plane, chunk)
                if (tap < 8) {
                    stage_b(wbase, tap + 1, kc, bbuf ^ 1);
                } else {
                    int nid = id, nkc = kc + 1;
                    if (nkc == nk) {
                        nkc = 0;
                        for (nid = id + 1; nid < cl.nD; ++nid)
                            if ((unsigned)(qd * p.mulD + cl.offD[nid]) < (unsigned)p.Ds) break;
                    }
                    if (nid < cl.nD) stage_b(nid == id ? wbase : wbase_of(nid), 0, nkc, bbuf ^ 1);
                }
                const int shift = cl.offH[ih] * PW + cl.offW[iw];
                const char *lb = bst + bbuf * B_BYTES + wn * (64 * 128);
                bool x3_done = false;
                if constexpr (F32) {
                    if (p.x3) {
                        if (tap == 0 && p.x3 != 3) {      // the patch has landed in every wave: split it once for all nine taps
                            x3_split_patch<256>(patch, PA * 32);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                        bf16x8_t bh[4], bl[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) x3_weights(p.x3 >= 2, lb + j * 2048 + x3_b0, lb + j * 2048 + x3_b1, bh[j], bl[j]);
#pragma unroll
                        for (int i = 0; i < SM; ++i) {
                            const int q = q0[i] + shift;
                            const bf16x8_t ah = *reinterpret_cast<const bf16x8_t *>(patch + q * 128 + (((2 * fq) ^ ((q >> 1) & 7)) << 4));
                            const bf16x8_t al = *reinterpret_cast<const bf16x8_t *>(patch + q * 128 + (((2 * fq + 1) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] = vn_mfma_x3(ah, al, bh[j], bl[j], acc[i][j]);
                        }
                        x3_done = true;
                    }
                }
                if (!x3_done) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int fo = ks ? bfrag1 : bfrag0;
                    if constexpr (F32) {
                        f32x4_t bq[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const f32x4_t *>(lb + j * 2048 + fo);
#pragma unroll
                        for (int i = 0; i < SM; ++i) {
                            const int q = q0[i] + shift;
                            const f32x4_t a = *reinterpret_cast<const f32x4_t *>(patch + q * 128 + (((ks * 4 + fq) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t4], bq[j][t4], acc[i][j], 0, 0, 0);
                        }
                    } else {
                        bf16x8_t bq[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bf16x8_t *>(lb + j * 2048 + fo);
#pragma unroll
                        for (int i = 0; i < SM; ++i) {
                            const int q = q0[i] + shift;
                            const bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(patch + q * 128 + (((ks * 4 + fq) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[j], acc[i][j], 0, 0, 0);
                        }
                    }
                }
                }
                bbuf ^= 1;
            }
        }
    }

    // ---- epilogue: row -> output offset table, then the shared store / statistics code
    __syncthreads();
    int32_t *otab = reinterpret_cast<int32_t *>(smem);
    for (int r = threadIdx.x; r < BM; r += 256) {
        const int py = r / TW, px = r - py * TW;
        const int qh = y0 + py, qw = x0 + px;
        int32_t off = -1;
        if (qh < cl.qH && qw < cl.qW) {
            const int od = qd * p.omulD + cl.ooffD, oh = qh * p.omulH + cl.ooffH, ow = qw * p.omulW + cl.ooffW;
            if (od < p.Do && oh < p.Ho && ow < p.Wo)
                off = (int32_t)((int64_t)b * p.oB + (int64_t)od * p.oD + (int64_t)oh * p.oH + (int64_t)ow * p.oW);
        }
        otab[r] = off;
    }
    __syncthreads();
    gg_store<WM, WN, SM>(p, acc, smem, otab, tile, n0, wm, wn, lane);
}

// 2-D layers on small images (100 x 88, 50 x 44): the same kernel with an NSB-deep weight pipeline.
// NW = WM x WN waves, 4 or 8.  These launches are 140-280 workgroups on 256 CUs: ONE workgroup per CU, and with four
// waves one wave per SIMD that serialises, per tap step, the issue of its LDS-DMA pieces (~113 clk per 1-KiB piece:
// four weight pieces = 450 clk), its fragment reads, 16 MFMAs (256 clk) and the barrier — ~1000 clk for 256 clk of
// MFMA, 36 times.  Eight waves (4 x 2 waves of 16 x 64) halve every wave's share of all three and put two waves on
// each SIMD, so that one wave's DMA issue runs beside the other's MFMAs (round 3).
template <int WM, int WN, int SM, int TW, int NSB, bool F32>
__global__ void __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) k_conv_patch2d(const GGParams p) {
    VN_PRIO_MAIN();
    constexpr int ESZ = F32 ? 4 : 2;
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 16 * SM * WM, BN = 64 * WN, TH = BM / TW;
    static_assert((NW == 4 || NW == 8) && BM % TW == 0 && (BN / 8) % NW == 0, "4 or 8 waves; whole patch lines");
    constexpr int PW = TW + 2, PH = TH + 2, PROWS = PH * PW;
    constexpr int PPIECES = (PROWS + 7) / 8;                 // 1-KiB pieces of the patch
    constexpr int PA = (PPIECES + NW - 1) / NW;              // per wave
    constexpr int PATCH_BYTES = PA * NW * 1024;
    constexpr int RB = BN / (8 * NW);                        // weight pieces per wave and tap
    constexpr int B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *bst = smem + 2 * PATCH_BYTES;                      // two patch buffers, then NSB weight stages
    static_assert(NSB >= 3 && (NSB - 2) * RB <= 63, "vmcnt is 6 bits");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const GGClass &cl = p.cls[blockIdx.y];
    const int ntn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tile = bid / ntn, tile_n = bid - tile * ntn;
    const int n0 = tile_n * BN;
    const int tiles_x = (cl.qW + TW - 1) / TW, tiles_y = (cl.qH + TH - 1) / TH;
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int qd = t % cl.qD;
    const int b = t / cl.qD;
    if (b >= p.B) return;                                    // (a residue class with fewer planes than the largest)
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- loader state: patch rows of this lane (fixed over the planes and K chunks) and its weight rows
    const int a_chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    uint32_t a_row[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int q = (i * NW + wave) * 8 + (lane >> 3);      // patch row
        const int qy = q / PW, qx = q - qy * PW;
        const int sy = y0 - 1 + qy, sx = x0 - 1 + qx;          // H/W: stride 1, offsets -1..1 (checked on the host)
        a_row[i] = (q < PROWS && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
                       ? (uint32_t)(((int64_t)sy * p.sH + (int64_t)sx * p.sW) * ESZ)
                       : GG_OOB;
    }
    uint32_t b_row[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int rho = (i * NW + wave) * 8 + (lane >> 3);
        const int rl = rho & 63;
        const int n = n0 + (rho & ~63) + (rl & 15) * 4 + (rl >> 4);
        b_row[i] = n < p.N ? (uint32_t)((int64_t)n * p.Cs * ESZ) : GG_OOB;
    }
    const char *src_base = p.src + (int64_t)b * p.sB * ESZ;
    int64_t src_bytes = p.src_batch_extent * ESZ;
    if (src_bytes > (int64_t)GG_MAX_WINDOW) src_bytes = GG_MAX_WINDOW;
    const __amdgpu_buffer_rsrc_t rs_a = vn_uniform_rsrc(src_base, (uint32_t)src_bytes);
    const __amdgpu_buffer_rsrc_t rs_b = vn_uniform_rsrc(p.w, p.w_bytes);
    constexpr int BKE = 128 / ESZ, EPC = 16 / ESZ;
    const int nk = (p.Cs + BKE - 1) / BKE;
    const int k_lane0 = a_chunk * EPC;

    // ---- fragment geometry: MFMA row (lane & 15) of sub-tile i is output pixel (py, px) -> patch row of tap (0,0)
    const int fr = lane & 15, fq = lane >> 4;
    int q0[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int r = wm * (16 * SM) + i * 16 + fr;
        const int py = r / TW, px = r - py * TW;
        q0[i] = (py + 1) * PW + (px + 1);
    }
    const int bfrag0 = fr * 128 + (((0 + fq) ^ (fr >> 1)) << 4);
    const int bfrag1 = fr * 128 + (((4 + fq) ^ (fr >> 1)) << 4);
    const int x3_b0 = fr * 128 + (((2 * fq) ^ (fr >> 1)) << 4);        // fp32x3: granules 2 fq (hi) / 2 fq + 1 (lo)
    const int x3_b1 = fr * 128 + (((2 * fq + 1) ^ (fr >> 1)) << 4);

    f32x4_t acc[SM][4];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // H/W have one class and their taps in kernel order (div 1), so the weight index of tap (id, ih, iw) is that of
    // (id, 0, 0) plus ih*3 + iw: one scalar load per depth tap, none inside the tap loop
    const uint32_t tap_bytes = (uint32_t)p.N * (uint32_t)p.Cs * (uint32_t)ESZ;
    auto wbase_of = [&](int id) { return p.taps[cl.tap_begin + id * 9].widx; };
    auto stage_b = [&](int wbase, int tap, int kc, int buf) {
        const int k_lane = kc * BKE + k_lane0;
        const bool k_ok = k_lane < p.Cs;
        const uint32_t b_koff = (uint32_t)k_lane * (uint32_t)ESZ;
        const uint32_t b_soff = __builtin_amdgcn_readfirstlane((uint32_t)(wbase + tap) * tap_bytes);
        char *lb = bst + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < RB; ++i)
            lds_dma16(rs_b, lb + i * (NW * 1024), (k_ok && b_row[i] != GG_OOB) ? b_row[i] + b_koff : GG_OOB, b_soff);
    };
    auto stage_patch = [&](int sd, int kc, char *patch) {
        const int k_lane = kc * BKE + k_lane0;
        const bool k_ok = k_lane < p.Cs;
        const int k_src = k_lane;
        const uint32_t off = (uint32_t)(((int64_t)sd * p.sD + k_src) * ESZ);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            lds_dma16(rs_a, patch + (i * NW + wave) * 1024, (k_ok && a_row[i] != GG_OOB) ? a_row[i] + off : GG_OOB, 0);
    };

    // one source plane (2-D layer): steps s = kc*9 + tap.  The weight tile of step s + NSB - 1 is staged while step s is on
    // the MFMAs and the patch of the next K chunk PT taps into this one (second patch buffer), so that bytes fetched from
    // HBM / Infinity Cache — in the train step every layer's weights and input are cold — have NSB - 1 tap steps to arrive
    // instead of one, and a chunk boundary is an ordinary step (no drain)
    constexpr int PT = 3;
#ifdef VN_P2D_TRACE
    const bool tr_on = blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0;
#endif
    const int sd = qd * p.mulD + cl.offD[0];
    const int total = ((unsigned)sd < (unsigned)p.Ds) ? nk * 9 : 0;
    const int wbase = wbase_of(0);
    auto kc_of = [&](int kc) { return kc; };
    auto stage_step = [&](int s) {
        const int kc = s / 9;
        stage_b(wbase, s - kc * 9, kc_of(kc), s % NSB);
    };
    if (total > 0) stage_patch(sd, kc_of(0), smem);
#pragma unroll
    for (int s = 0; s < NSB - 1; ++s)
        if (s < total) stage_step(s);
    int pbuf = 0;
    for (int kc = 0; kc < nk && total > 0; ++kc) {
        const char *patch = smem + pbuf * PATCH_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = kc * 9 + tap;
            const int ih = tap / 3, iw = tap - ih * 3;
            // the tile of step s (and, at tap 0, this chunk's patch: issued before that tile) has landed; the loads issued
            // after it — the tiles of steps s+1 .. s+NSB-2 — may still be in flight.  lgkmcnt(0): see k_gather_gemm
            const int after = total - 1 - s < NSB - 2 ? total - 1 - s : NSB - 2;
            P2D_STAMP(s, 0);
            if (after >= 2 && NSB >= 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * RB) : "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            P2D_STAMP(s, 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            P2D_STAMP(s, 2);
            // (the other patch buffer was last read in the previous chunk: PT >= 1 barriers ago)
            if (tap == PT && kc + 1 < nk) stage_patch(sd, kc_of(kc + 1), smem + (pbuf ^ 1) * PATCH_BYTES);
            if (s + NSB - 1 < total) stage_step(s + NSB - 1);
            P2D_STAMP(s, 3);
            const int shift = cl.offH[ih] * PW + cl.offW[iw];
            const char *lb = bst + (s % NSB) * B_BYTES + wn * (64 * 128);
            bool x3_done = false;
            if constexpr (F32) {
                if (p.x3) {
                    if (tap == 0 && p.x3 != 3) {      // this chunk's patch has landed in every wave: split it once for all nine taps
                        x3_split_patch<NT>(smem + pbuf * PATCH_BYTES, PA * NW * 8);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    bf16x8_t bh[4], bl[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x3_weights(p.x3 >= 2, lb + j * 2048 + x3_b0, lb + j * 2048 + x3_b1, bh[j], bl[j]);
#pragma unroll
                    for (int i = 0; i < SM; ++i) {
                        const int q = q0[i] + shift;
                        const bf16x8_t ah = *reinterpret_cast<const bf16x8_t *>(patch + q * 128 + (((2 * fq) ^ ((q >> 1) & 7)) << 4));
                        const bf16x8_t al = *reinterpret_cast<const bf16x8_t *>(patch + q * 128 + (((2 * fq + 1) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = vn_mfma_x3(ah, al, bh[j], bl[j], acc[i][j]);
                    }
                    x3_done = true;
                }
            }
            if (!x3_done) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int fo = ks ? bfrag1 : bfrag0;
                if constexpr (F32) {
                    f32x4_t bq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const f32x4_t *>(lb + j * 2048 + fo);
#pragma unroll
                    for (int i = 0; i < SM; ++i) {
                        const int q = q0[i] + shift;
                        const f32x4_t a = *reinterpret_cast<const f32x4_t *>(patch + q * 128 + (((ks * 4 + fq) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t4], bq[j][t4], acc[i][j], 0, 0, 0);
                    }
                } else {
                    bf16x8_t bq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bf16x8_t *>(lb + j * 2048 + fo);
#pragma unroll
                    for (int i = 0; i < SM; ++i) {
                        const int q = q0[i] + shift;
                        const bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(patch + q * 128 + (((ks * 4 + fq) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            }
#ifdef VN_P2D_TRACE
            if (tr_on) {     // wait for this step's accumulators (adds a dependency the product build does not have)
                asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[SM - 1][3]));
                P2D_STAMP(s, 4);
            }
#endif
        }
        pbuf ^= 1;
    }

    // ---- epilogue: row -> output offset table, then the shared store / statistics code
    __syncthreads();
    int32_t *otab = reinterpret_cast<int32_t *>(smem);
    for (int r = threadIdx.x; r < BM; r += NT) {
        const int py = r / TW, px = r - py * TW;
        const int qh = y0 + py, qw = x0 + px;
        int32_t off = -1;
        if (qh < cl.qH && qw < cl.qW) {
            const int od = qd * p.omulD + cl.ooffD, oh = qh * p.omulH + cl.ooffH, ow = qw * p.omulW + cl.ooffW;
            if (od < p.Do && oh < p.Ho && ow < p.Wo)
                off = (int32_t)((int64_t)b * p.oB + (int64_t)od * p.oD + (int64_t)oh * p.oH + (int64_t)ow * p.oW);
        }
        otab[r] = off;
    }
    __syncthreads();
    if (p.bn_y) gg_store<WM, WN, SM, true>(p, acc, smem, otab, tile, n0, wm, wn, lane);
    else gg_store<WM, WN, SM>(p, acc, smem, otab, tile, n0, wm, wn, lane);
}

inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int posmod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

struct AxisClass {
    int r;          // residue (out offset)
    int q;          // row-grid size along this axis
    int n;          // taps in this class
    int tap[4];     // kernel index
    int off[4];     // source coordinate offset
};

// decompose one axis: s = (row*mul + t*tmul - pad)/div.  Returns number of classes.
int axis_classes(int R, int k, int mul, int tmul, int pad, int div, AxisClass *out, int *mul_out, int *omul_out) {
    if (div == 1) {
        AxisClass c{};
        c.r = 0; c.q = R; c.n = k;
        for (int t = 0; t < k; ++t) { c.tap[t] = t; c.off[t] = t * tmul - pad; }
        out[0] = c;
        *mul_out = mul; *omul_out = 1;
        return 1;
    }
    // mul must be 1: row = div*q + r
    int nc = 0;
    for (int r = 0; r < div && r < R; ++r) {
        AxisClass c{};
        c.r = r; c.q = (R - r + div - 1) / div; c.n = 0;
        for (int t = 0; t < k; ++t) {
            const int num = r + t * tmul - pad;
            if (posmod(num, div) == 0) { c.tap[c.n] = t; c.off[c.n] = floordiv(num, div); ++c.n; }
        }
        out[nc++] = c;
    }
    *mul_out = 1; *omul_out = div;
    return nc;
}

// the dynamic-LDS attribute is set once per instantiation (function-local static: not inside a stream capture)
template <int WM, int WN, int SM, int NS, bool F32>
int launch_gg(const GGParams &p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * (16 * SM * WM + 64 * WN) * 128;
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void *>(&k_gather_gemm<WM, WN, SM, NS, F32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    k_gather_gemm<WM, WN, SM, NS, F32><<<grid, 256, lds, st>>>(p);
    return 0;
}

// Tile configurations.  The two big ones keep a 64x64 tile per wave (best LDS bytes per MFMA); the small ones trade
// that for enough workgroups to cover 256 CUs and for a deep load pipeline (a layer with 4400 output rows is 70
// 128x128 tiles: one latency-bound workgroup on a quarter of the chip).
struct GGConfig {
    int id, BM, BN;
};
constexpr GGConfig GG_CFG[5] = {{0, 256, 64}, {1, 128, 128}, {2, 64, 128}, {3, 64, 64}, {4, 160, 128}};

int gg_force() {   // tuning aid: VN_GG_CONFIG=0..4 forces one configuration
    static const int v = vn_knob("VN_GG_CONFIG", -1);
    return v;
}

GGConfig gg_config(int64_t rows, int Cr, int ncls) {
    const int f = gg_force();
    if (f >= 0 && f < 5) return GG_CFG[f];
    auto blocks = [&](const GGConfig &c) { return vn_ceil_div(rows, c.BM) * vn_ceil_div(Cr, c.BN) * ncls; };
    if (Cr <= 64) return GG_CFG[0];
    // 128x128 and 160x128 both run two workgroups per CU (512 slots): take the one whose last round of workgroups
    // is fuller — 70,400 rows are 550 tiles of 128 (a second round of 38) but 440 tiles of 160 (one round)
    const int64_t b1 = blocks(GG_CFG[1]), b4 = blocks(GG_CFG[4]);
    if (b1 < 100) return GG_CFG[2];              // a handful of long-K tiles: smaller tiles, 4-stage pipeline
    const int64_t t1 = vn_ceil_div(b1, 512) * 128, t4 = vn_ceil_div(b4, 512) * 160;
    return t4 < t1 ? GG_CFG[4] : GG_CFG[1];
}

// ---- patch kernel (k_conv_patch) eligibility and tiling
struct PatchCfg {
    int id, BM, BN, TW;   // id < 0: not eligible
};
int patch_enabled() {     // tuning aid: VN_PATCH=0 keeps every layer on k_gather_gemm, 2 = also small images
    static const int v = vn_knob("VN_PATCH", 1);
    return v;
}
PatchCfg patch_config(const vnConv *g) {
    PatchCfg none{-1, 0, 0, 0};
    if (!patch_enabled()) return none;
    if (g->kH != 3 || g->kW != 3 || g->mulH != 1 || g->mulW != 1 || g->divH != 1 || g->divW != 1) return none;
    // source offsets t*tmul - pad of the three taps must be {-1, 0, 1}
    if (!((g->tmulH == 1 && g->padH == 1) || (g->tmulH == -1 && g->padH == -1))) return none;
    if (!((g->tmulW == 1 && g->padW == 1) || (g->tmulW == -1 && g->padW == -1))) return none;
    if (g->src_wrap != 0 || (g->Cr & 63)) return none;
    if (g->Hs != g->Hr || g->Ws != g->Wr) return none;
    if (g->Hr < 128 || g->Wr < 128) {
        // small images (100 x 88, 50 x 44): 4 x 16-pixel tiles so that there are enough workgroups for 256 CUs
        // (VN_PATCH=2: every eligible layer, for tests; VN_PATCH=3: small images on k_gather_gemm as before)
        if (patch_enabled() == 3 || g->Cr == 64) return none;
        return PatchCfg{2, 64, 128, 16};
    }
    // 64-channel Conv3d layers: 6 x 32 pixels (patch 35 KB + 2 weight stages = 51 KB: THREE workgroups per CU; measured
    // 1340 vs 1183 TFLOP/s on middle_layer.2 against 8 x 32 pixels / two workgroups per CU)
    if (g->Cr == 64) return PatchCfg{3, 192, 64, 32};
    return PatchCfg{0, 160, 128, 16};                        // 10 x 16 pixels
}
int64_t patch_tiles(const PatchCfg &c, int B, int qD, int qH, int qW) {
    const int TH = c.BM / c.TW;
    return (int64_t)B * qD * vn_ceil_div(qH, TH) * vn_ceil_div(qW, c.TW);
}
template <int WM, int WN, int SM, int TW, bool F32>
int launch_patch(const GGParams &p, dim3 grid, hipStream_t st) {
    constexpr int BM = 16 * SM * WM, TH = BM / TW, PROWS = (TH + 2) * (TW + 2);
    constexpr size_t lds = (size_t)((PROWS + 7) / 8 + 3) / 4 * 4096 + 2 * (size_t)(64 * WN) * 128;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_patch<WM, WN, SM, TW, F32>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    k_conv_patch<WM, WN, SM, TW, F32><<<grid, 256, lds, st>>>(p);
    return 0;
}
template <int WM, int WN, int SM, int TW, int NSB, bool F32>
int launch_patch2d(const GGParams &p, dim3 grid, hipStream_t st) {
    constexpr int BM = 16 * SM * WM, TH = BM / TW, PROWS = (TH + 2) * (TW + 2), NW = WM * WN;
    constexpr size_t lds = 2 * ((size_t)((PROWS + 7) / 8 + NW - 1) / NW * NW * 1024) + NSB * (size_t)(64 * WN) * 128;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_patch2d<WM, WN, SM, TW, NSB, F32>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    k_conv_patch2d<WM, WN, SM, TW, NSB, F32><<<grid, 64 * NW, lds, st>>>(p);
    return 0;
}
int patch2d_stages() {   // VN_PATCH2D=0: the two-stage kernel for the small images too; 3 / 4: weight stages
    static const int v = vn_knob("VN_PATCH2D", 3);
    return v;
}
int launch_patch_cfg(const PatchCfg &c, bool f32, const GGParams &p, dim3 grid, hipStream_t st) {
    if (c.id == 2 && p.Ds == 1 && p.nclasses == 1 && p.cls[0].nD == 1 && patch2d_stages() >= 3) {
        const GGParams &q = p;
        // eight waves (4 x 2 of 16 x 64) for the bf16 kernels (the four-wave kernel of round 2 stays for the exact fp32 path)
        if (patch2d_stages() == 3 && !f32) return launch_patch2d<4, 2, 1, 16, 3, false>(q, grid, st);
        // fp32x3: with the splits out of the tap loop its steps look like the bf16 kernel's (DMA issue, fragment reads, 24 bf16
        // MFMAs per 16 x 64 sub-tile) — the same eight-wave arrangement; the exact fp32 path (128 fp32 MFMAs per step) keeps four
        if (patch2d_stages() == 3 && f32 && q.x3) return launch_patch2d<4, 2, 1, 16, 3, true>(q, grid, st);
        if (patch2d_stages() == 3)
            return f32 ? launch_patch2d<2, 2, 2, 16, 3, true>(q, grid, st) : launch_patch2d<2, 2, 2, 16, 3, false>(q, grid, st);
        return f32 ? launch_patch2d<2, 2, 2, 16, 4, true>(q, grid, st) : launch_patch2d<2, 2, 2, 16, 4, false>(q, grid, st);
    }
    if (c.id == 3) return f32 ? launch_patch<4, 1, 3, 32, true>(p, grid, st) : launch_patch<4, 1, 3, 32, false>(p, grid, st);
    if (c.id == 2) return f32 ? launch_patch<2, 2, 2, 16, true>(p, grid, st) : launch_patch<2, 2, 2, 16, false>(p, grid, st);
    return f32 ? launch_patch<2, 2, 5, 16, true>(p, grid, st) : launch_patch<2, 2, 5, 16, false>(p, grid, st);
}

int launch_cfg(const GGConfig &c, bool f32, const GGParams &p, dim3 grid, hipStream_t st) {
    switch (c.id) {
    case 0: return f32 ? launch_gg<4, 1, 4, 2, true>(p, grid, st) : launch_gg<4, 1, 4, 2, false>(p, grid, st);
    case 1: return f32 ? launch_gg<2, 2, 4, 2, true>(p, grid, st) : launch_gg<2, 2, 4, 2, false>(p, grid, st);
    case 2: return f32 ? launch_gg<2, 2, 2, 4, true>(p, grid, st) : launch_gg<2, 2, 2, 4, false>(p, grid, st);
    case 3: return f32 ? launch_gg<4, 1, 1, 6, true>(p, grid, st) : launch_gg<4, 1, 1, 6, false>(p, grid, st);
    default: return f32 ? launch_gg<2, 2, 5, 2, true>(p, grid, st) : launch_gg<2, 2, 5, 2, false>(p, grid, st);
    }
}

// list-mode slots of one axis: every (residue r, tap t) with (r + t*tmul - pad) divisible by div
struct AxisSlots {
    int n, cls[4], off[4], tap[4];
};
bool axis_slots(int k, int tmul, int pad, int div, AxisSlots *o) {
    o->n = 0;
    for (int r = 0; r < div; ++r)
        for (int t = 0; t < k; ++t) {
            const int num = r + t * tmul - pad;
            if (posmod(num, div) != 0) continue;
            if (o->n >= 4) return false;
            o->cls[o->n] = div == 1 ? -1 : r;
            o->off[o->n] = floordiv(num, div);
            o->tap[o->n] = t;
            ++o->n;
        }
    return o->n > 0;
}

}  // namespace

extern "C" int vn_conv_gather_gemm_rows(const void *src, const void *w_packed, const float *bias, void *out,
                                        vnDtype out_dtype, const vnConv *g, const int64_t *row_list, int64_t row_cap,
                                        const int32_t *row_count, int32_t out_linear, float *stats_slab,
                                        vnStream stream) {
    VN_CHECK_ARG(src && w_packed && out && g && row_list && row_cap >= 0 && row_cap < (1ll << 31));
    if (row_cap == 0) return VN_OK;
    VN_CHECK_ARG(g->B > 0 && g->Ds > 0 && g->Hs > 0 && g->Ws > 0 && g->Dr > 0 && g->Hr > 0 && g->Wr > 0);
    VN_CHECK_ARG(g->kD >= 1 && g->kD <= 4 && g->kH >= 1 && g->kH <= 4 && g->kW >= 1 && g->kW <= 4);
    VN_CHECK_ARG(g->divD >= 1 && g->divH >= 1 && g->divW >= 1);
    VN_CHECK_ARG((g->divD == 1 || g->mulD == 1) && (g->divH == 1 || g->mulH == 1) && (g->divW == 1 || g->mulW == 1));
    VN_CHECK_ARG(out_dtype == VN_F32 || out_dtype == VN_BF16);
    VN_CHECK_ARG(g->dtype == VN_BF16 || g->dtype == VN_F32 || g->dtype == VN_F32X3 || g->dtype == VN_F32X3S);
    const bool f32 = g->dtype != VN_BF16;      // VN_F32X3: the fp32 kernels (fp32 storage) with the three-bf16-product inner loop
    const int esz = f32 ? 4 : 2, align_e = 16 / esz;
    if (g->Cs <= 0 || (g->Cs % align_e) || g->Cr <= 0 || (g->Cr & 3) || g->src_wrap != 0) return VN_EUNSUPPORTED;
    if (((g->src_sB | g->src_sD | g->src_sH | g->src_sW) & (align_e - 1)) != 0) return VN_EUNSUPPORTED;
    if (((g->out_sB | g->out_sD | g->out_sH | g->out_sW) & 3) != 0) return VN_EUNSUPPORTED;
    GGParams p{};
    p.src = static_cast<const char *>(src);
    p.w = static_cast<const char *>(w_packed);
    p.bias = bias;
    p.out = static_cast<char *>(out);
    p.stats = stats_slab;
    p.sB = g->src_sB; p.sD = g->src_sD; p.sH = g->src_sH; p.sW = g->src_sW;
    p.oB = g->out_sB; p.oD = g->out_sD; p.oH = g->out_sH; p.oW = g->out_sW;
    p.B = g->B; p.Ds = g->Ds; p.Hs = g->Hs; p.Ws = g->Ws;
    p.Do = g->Dr; p.Ho = g->Hr; p.Wo = g->Wr;
    p.Cs = g->Cs; p.src_wrap = 0; p.N = g->Cr;
    p.out_f32 = out_dtype == VN_F32;
    p.src_row_elems = g->Cs;
    p.esz = esz;
    p.x3 = g->dtype == VN_F32X3 ? (vn_x3_presplit(g->Cs) ? 2 : 1) : 0;   // 2: weights split by vn_pack_weight (VN_F32X3 operand)
    if (g->dtype == VN_F32X3S) {      // the source rows are stored split too (BatchNorm passes: VN_F32X3S tensors)
        if (!vn_x3_presplit(g->Cs) || ((g->src_sB | g->src_sD | g->src_sH | g->src_sW) & 7)) return VN_EUNSUPPORTED;
        p.x3 = 3;
    }
    const int taps_total = g->kD * g->kH * g->kW;
    const int64_t wb = (int64_t)taps_total * g->Cr * g->Cs * esz;
    if (wb > (int64_t)GG_MAX_WINDOW) return VN_EUNSUPPORTED;
    p.w_bytes = (uint32_t)wb;
    p.src_batch_extent = (int64_t)(g->Ds - 1) * g->src_sD + (int64_t)(g->Hs - 1) * g->src_sH +
                         (int64_t)(g->Ws - 1) * g->src_sW + g->Cs;
    // list rows may come from any batch item: the whole source tensor must fit one 32-bit window
    if (((int64_t)(g->B - 1) * g->src_sB + p.src_batch_extent) * esz > (int64_t)GG_MAX_WINDOW - 4096) return VN_EUNSUPPORTED;
    const int64_t out_extent = out_linear ? row_cap * g->out_sW
                                          : (int64_t)(g->B - 1) * g->out_sB + (int64_t)(g->Dr - 1) * g->out_sD +
                                                (int64_t)(g->Hr - 1) * g->out_sH + (int64_t)(g->Wr - 1) * g->out_sW + g->Cr;
    if (out_extent >= (1ll << 31)) return VN_EUNSUPPORTED;
    AxisSlots sd, sh, sw;
    if (!axis_slots(g->kD, g->tmulD, g->padD, g->divD, &sd) || !axis_slots(g->kH, g->tmulH, g->padH, g->divH, &sh) ||
        !axis_slots(g->kW, g->tmulW, g->padW, g->divW, &sw))
        return VN_EUNSUPPORTED;
    if (sd.n * sh.n * sw.n > GG_MAX_TAPS) return VN_EUNSUPPORTED;
    p.row_list = row_list;
    p.row_count = row_count;
    p.row_cap = (int32_t)row_cap;
    p.out_linear = out_linear;
    p.ldivD = g->divD; p.ldivH = g->divH; p.ldivW = g->divW;
    p.mulD = g->divD == 1 ? g->mulD : 1;
    p.mulH = g->divH == 1 ? g->mulH : 1;
    p.mulW = g->divW == 1 ? g->mulW : 1;
    p.omulD = p.omulH = p.omulW = 1;
    GGClass &c = p.cls[0];
    c.tap_begin = 0;
    c.nD = sd.n; c.nH = sh.n; c.nW = sw.n;
    c.qD = c.qH = c.qW = 1;
    c.ooffD = c.ooffH = c.ooffW = 0;
    for (int j = 0; j < 4; ++j) {
        c.offD[j] = j < sd.n ? sd.off[j] : (1 << 29);
        c.offH[j] = j < sh.n ? sh.off[j] : (1 << 29);
        c.offW[j] = j < sw.n ? sw.off[j] : (1 << 29);
        p.slotcD[j] = j < sd.n ? sd.cls[j] : -2;
        p.slotcH[j] = j < sh.n ? sh.cls[j] : -2;
        p.slotcW[j] = j < sw.n ? sw.cls[j] : -2;
    }
    int ntap = 0;
    for (int i = 0; i < sd.n; ++i)
        for (int j = 0; j < sh.n; ++j)
            for (int k = 0; k < sw.n; ++k)
                p.taps[ntap++] = GGTap{i, j, k, (sd.tap[i] * g->kH + sh.tap[j]) * g->kW + sw.tap[k]};
    c.ntaps = ntap;
    p.nclasses = 1;
    // capacity launch: rows are not known on the host
    const GGConfig cfg = g->Cr > 64 ? GG_CFG[1] : GG_CFG[0];
    const int64_t tiles_m = vn_ceil_div(row_cap, cfg.BM), tiles_n = vn_ceil_div(g->Cr, cfg.BN);
    const dim3 grid((unsigned)(tiles_m * tiles_n), 1);
    hipStream_t st = vn_stream(stream);
    const int rc = launch_cfg(cfg, f32, p, grid, st);
    if (rc) return rc;
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int64_t vn_conv_stats_slab_rows(const vnConv *g) {
    if (!g || g->divD < 1 || g->divH < 1 || g->divW < 1) return 0;
    if (g->divD * g->divH * g->divW > 1) {
        // residue classes (strided transposed gathers): every class launches the tiles of the largest one
        const int64_t qd = vn_ceil_div(g->Dr, g->divD), qh = vn_ceil_div(g->Hr, g->divH), qw = vn_ceil_div(g->Wr, g->divW);
        const int nd = g->divD < g->Dr ? g->divD : g->Dr, nh = g->divH < g->Hr ? g->divH : g->Hr, nw = g->divW < g->Wr ? g->divW : g->Wr;
        const int64_t rows = (int64_t)g->B * qd * qh * qw;
        const int ncls = nd * nh * nw;
        return (int64_t)ncls * vn_ceil_div(rows, gg_config(rows, g->Cr, ncls).BM);
    }
    const PatchCfg pc = patch_config(g);
    if (pc.id >= 0) return patch_tiles(pc, g->B, g->Dr, g->Hr, g->Wr);
    const int64_t rows = (int64_t)g->B * g->Dr * g->Hr * g->Wr;
    return vn_ceil_div(rows, gg_config(rows, g->Cr, 1).BM);
}

// Which kernel / tile vn_conv_gather_gemm picks for a geometry (tests assert that the production instantiations are
// the ones compared with the oracle): 100 + patch tile id (k_conv_patch: 0 = 10x16, 1 = 8x32, 2 = 4x16, 3 = 6x32,
// 4 = 6x16, 5 = 8x16 pixels), 120 + weight stages (k_conv_patch2d, 4x16 pixels) or the k_gather_gemm tile id (0 = 256x64,
// 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 160x128).
extern "C" int32_t vn_conv_plan_id(const vnConv *g) {
    if (!g || g->divD < 1 || g->divH < 1 || g->divW < 1 || g->B <= 0) return -1;
    const PatchCfg pc = patch_config(g);          // (depth residue classes are fine for the patch kernel)
    if (pc.id == 2 && g->Ds == 1 && g->Dr == 1 && g->kD == 1 && g->divD == 1 && patch2d_stages() >= 3)
        return 120 + patch2d_stages();            // k_conv_patch2d (4x16 pixels, deep weight pipeline): 123
    if (pc.id >= 0) return 100 + pc.id;
    const int64_t qd = vn_ceil_div(g->Dr, g->divD), qh = vn_ceil_div(g->Hr, g->divH), qw = vn_ceil_div(g->Wr, g->divW);
    const int nd = g->divD < g->Dr ? g->divD : g->Dr, nh = g->divH < g->Hr ? g->divH : g->Hr, nw = g->divW < g->Wr ? g->divW : g->Wr;
    return gg_config((int64_t)g->B * qd * qh * qw, g->Cr, nd * nh * nw).id;   // largest class x number of classes
}

static int gather_gemm_impl(const void *src, const void *w_packed, const float *bias, void *out, vnDtype out_dtype,
                            const vnConv *g, int32_t accumulate, float *stats_slab, vnStream stream, const void *bn_y,
                            vnDtype bn_y_dtype, const float *bn_stats);

extern "C" int vn_conv_gather_gemm(const void *src, const void *w_packed, const float *bias, void *out,
                                   vnDtype out_dtype, const vnConv *g, int32_t accumulate, float *stats_slab,
                                   vnStream stream) {
    return gather_gemm_impl(src, w_packed, bias, out, out_dtype, g, accumulate, stats_slab, stream, nullptr, VN_BF16, nullptr);
}

// A data-gradient launch that also leaves the BatchNorm-BACKWARD sums of the layer BELOW in its epilogue: out = the
// gradient w.r.t. that layer's activation a = relu(BN(y)); slab[vn_conv_stats_slab_rows(g)][2][Cr] receives, per workgroup,
// sum dz and sum dz*xhat (dz = the relu-masked out value as stored) — the rows vn_bn_bwd_finalize_slab reads, i.e. the
// vn_bn_bwd_reduce_slab launch (ConvMD backward, model.py:142-166) is saved.  Only for the geometries of the small-image
// 3x3 kernel (vn_conv_plan_id 123); VN_EUNSUPPORTED otherwise (use the two calls).
extern "C" int vn_conv_dgrad_bn_bwd(const void *src, const void *w_packed, void *out, vnDtype out_dtype, const vnConv *g,
                                    const void *bn_y, vnDtype bn_y_dtype, const float *bn_stats, float *slab,
                                    vnStream stream) {
    VN_CHECK_ARG(bn_y && bn_stats && slab && g);
    if (vn_conv_plan_id(g) != 123) return VN_EUNSUPPORTED;
    return gather_gemm_impl(src, w_packed, nullptr, out, out_dtype, g, 0, slab, stream, bn_y, bn_y_dtype, bn_stats);
}

static int gather_gemm_impl(const void *src, const void *w_packed, const float *bias, void *out, vnDtype out_dtype,
                            const vnConv *g, int32_t accumulate, float *stats_slab, vnStream stream, const void *bn_y,
                            vnDtype bn_y_dtype, const float *bn_stats) {
    VN_CHECK_ARG(src && w_packed && out && g);
    VN_CHECK_ARG(g->B > 0 && g->Ds > 0 && g->Hs > 0 && g->Ws > 0 && g->Dr > 0 && g->Hr > 0 && g->Wr > 0);
    VN_CHECK_ARG(g->kD >= 1 && g->kD <= 4 && g->kH >= 1 && g->kH <= 4 && g->kW >= 1 && g->kW <= 4);
    VN_CHECK_ARG(g->kD * g->kH * g->kW <= GG_MAX_TAPS);
    VN_CHECK_ARG(g->divD >= 1 && g->divH >= 1 && g->divW >= 1);
    VN_CHECK_ARG(g->divD == 1 || g->mulD == 1);
    VN_CHECK_ARG(g->divH == 1 || g->mulH == 1);
    VN_CHECK_ARG(g->divW == 1 || g->mulW == 1);
    VN_CHECK_ARG(out_dtype == VN_F32 || out_dtype == VN_BF16);
    VN_CHECK_ARG(!stats_slab || g->divD * g->divH * g->divW <= GG_MAX_CLASSES);
    VN_CHECK_ARG(g->dtype == VN_BF16 || g->dtype == VN_F32 || g->dtype == VN_F32X3 || g->dtype == VN_F32X3S);
    const bool f32 = g->dtype != VN_BF16;      // VN_F32X3: the fp32 kernels (fp32 storage) with the three-bf16-product inner loop
    const int esz = f32 ? 4 : 2, bke = 128 / esz, align_e = 16 / esz;
    (void)bke;
    if (g->Cs <= 0 || (g->Cs % align_e) || g->Cr <= 0 || (g->Cr & 3)) return VN_EUNSUPPORTED;
    if (g->src_wrap != 0) return VN_EUNSUPPORTED;      // (K wrap of [hi|lo] sources: the retired bf16x3 mode, round 5)
    if (g->divD * g->divH * g->divW > GG_MAX_CLASSES) return VN_EUNSUPPORTED;
    if (((g->src_sB | g->src_sD | g->src_sH | g->src_sW) & (align_e - 1)) != 0) return VN_EUNSUPPORTED;   // 16-B chunks
    if (((g->out_sB | g->out_sD | g->out_sH | g->out_sW) & 3) != 0) return VN_EUNSUPPORTED;      // 8/16-B stores
    if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return VN_EUNSUPPORTED;

    GGParams p{};
    p.src = static_cast<const char *>(src);
    p.w = static_cast<const char *>(w_packed);
    p.bias = bias;
    p.out = static_cast<char *>(out);
    p.stats = stats_slab;
    p.bn_y = bn_y;
    p.bn_stats = bn_stats;
    p.bn_y_f32 = bn_y_dtype == VN_F32;
    p.sB = g->src_sB; p.sD = g->src_sD; p.sH = g->src_sH; p.sW = g->src_sW;
    p.oB = g->out_sB; p.oD = g->out_sD; p.oH = g->out_sH; p.oW = g->out_sW;
    p.B = g->B; p.Ds = g->Ds; p.Hs = g->Hs; p.Ws = g->Ws;
    p.Do = g->Dr; p.Ho = g->Hr; p.Wo = g->Wr;
    p.Cs = g->Cs; p.src_wrap = g->src_wrap; p.N = g->Cr;
    p.out_f32 = out_dtype == VN_F32;
    p.accumulate = accumulate;
    p.src_row_elems = g->src_wrap > 0 ? g->src_wrap : g->Cs;
    p.esz = esz;
    p.x3 = g->dtype == VN_F32X3 ? (vn_x3_presplit(g->Cs) ? 2 : 1) : 0;   // 2: weights split by vn_pack_weight (VN_F32X3 operand)
    if (g->dtype == VN_F32X3S) {      // the source rows are stored split too (BatchNorm passes: VN_F32X3S tensors)
        if (!vn_x3_presplit(g->Cs) || ((g->src_sB | g->src_sD | g->src_sH | g->src_sW) & 7)) return VN_EUNSUPPORTED;
        p.x3 = 3;
    }
    const int taps_total = g->kD * g->kH * g->kW;
    const int64_t wb = (int64_t)taps_total * g->Cr * g->Cs * esz;
    if (wb > (int64_t)GG_MAX_WINDOW) return VN_EUNSUPPORTED;
    p.w_bytes = (uint32_t)wb;
    p.src_batch_extent = (int64_t)(g->Ds - 1) * g->src_sD + (int64_t)(g->Hs - 1) * g->src_sH +
                         (int64_t)(g->Ws - 1) * g->src_sW + p.src_row_elems;
    // output offsets are int32 elements
    const int64_t out_extent = (int64_t)(g->B - 1) * g->out_sB + (int64_t)(g->Dr - 1) * g->out_sD +
                               (int64_t)(g->Hr - 1) * g->out_sH + (int64_t)(g->Wr - 1) * g->out_sW + g->Cr;
    if (out_extent >= (1ll << 31)) return VN_EUNSUPPORTED;

    AxisClass ad[4], ah[4], aw[4];
    const int ncd = axis_classes(g->Dr, g->kD, g->mulD, g->tmulD, g->padD, g->divD, ad, &p.mulD, &p.omulD);
    const int nch = axis_classes(g->Hr, g->kH, g->mulH, g->tmulH, g->padH, g->divH, ah, &p.mulH, &p.omulH);
    const int ncw = axis_classes(g->Wr, g->kW, g->mulW, g->tmulW, g->padW, g->divW, aw, &p.mulW, &p.omulW);
    int ntap = 0, ncls = 0;
    int64_t max_rows = 0;
    for (int cd = 0; cd < ncd; ++cd)
        for (int ch = 0; ch < nch; ++ch)
            for (int cw = 0; cw < ncw; ++cw) {
                const AxisClass &D = ad[cd], &H = ah[ch], &W = aw[cw];
                GGClass &c = p.cls[ncls];
                c.tap_begin = ntap;
                c.nD = D.n; c.nH = H.n; c.nW = W.n;
                c.qD = D.q; c.qH = H.q; c.qW = W.q;
                c.ooffD = D.r; c.ooffH = H.r; c.ooffW = W.r;
                for (int j = 0; j < 4; ++j) {
                    c.offD[j] = j < D.n ? D.off[j] : (1 << 29);   // never valid
                    c.offH[j] = j < H.n ? H.off[j] : (1 << 29);
                    c.offW[j] = j < W.n ? W.off[j] : (1 << 29);
                }
                for (int i = 0; i < D.n; ++i)
                    for (int j = 0; j < H.n; ++j)
                        for (int k = 0; k < W.n; ++k) {
                            if (ntap >= GG_MAX_TAPS) return VN_EUNSUPPORTED;
                            p.taps[ntap++] = GGTap{i, j, k, (D.tap[i] * g->kH + H.tap[j]) * g->kW + W.tap[k]};
                        }
                c.ntaps = ntap - c.tap_begin;
                const int64_t rows = (int64_t)g->B * D.q * H.q * W.q;
                if (rows > max_rows) max_rows = rows;
                ++ncls;
            }
    p.nclasses = ncls;

    const PatchCfg pc = patch_config(g);
    if (pc.id >= 0) {
        // every class shares the H/W geometry (div 1 there); classes differ in depth only
        int64_t tiles = 0;
        for (int c = 0; c < ncls; ++c) {
            const int64_t t = patch_tiles(pc, g->B, p.cls[c].qD, p.cls[c].qH, p.cls[c].qW);
            if (t > tiles) tiles = t;
        }
        if (p.src_batch_extent * esz > (int64_t)GG_MAX_WINDOW - 4096) return VN_EUNSUPPORTED;
        const int64_t nblk = tiles * vn_ceil_div(g->Cr, pc.BN);
        if (nblk > 0x7fffffffll) return VN_EUNSUPPORTED;
        const dim3 pgrid((unsigned)nblk, (unsigned)ncls);
        const int prc = launch_patch_cfg(pc, f32, p, pgrid, vn_stream(stream));
        if (prc) return prc;
        VN_LAUNCH_STATUS();
        return VN_OK;
    }
    const GGConfig cfg = gg_config(max_rows, g->Cr, ncls);
    const int BM = cfg.BM, BN = cfg.BN;
    // the gather window of one workgroup: rows of at most (BM / rows_per_batch + 2) batch items
    {
        int64_t min_rows_b = (int64_t)p.cls[0].qD * p.cls[0].qH * p.cls[0].qW;
        for (int c = 1; c < ncls; ++c) {
            const int64_t r = (int64_t)p.cls[c].qD * p.cls[c].qH * p.cls[c].qW;
            if (r < min_rows_b) min_rows_b = r;
        }
        if (min_rows_b <= 0) min_rows_b = 1;
        int64_t span_b = BM / min_rows_b + 2;
        if (span_b > g->B) span_b = g->B;
        const int64_t need = ((span_b - 1) * g->src_sB + p.src_batch_extent) * esz;
        if (need > (int64_t)GG_MAX_WINDOW - 4096) return VN_EUNSUPPORTED;
    }
    const int64_t tiles_m = vn_ceil_div(max_rows, BM);
    const int64_t tiles_n = vn_ceil_div(g->Cr, BN);
    if (tiles_m * tiles_n > 0x7fffffffll) return VN_EUNSUPPORTED;
    const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)ncls);
    hipStream_t st = vn_stream(stream);
    const int rc = launch_cfg(cfg, f32, p, grid, st);
    if (rc) return rc;
    VN_LAUNCH_STATUS();
    return VN_OK;
}

#ifdef VN_P2D_TRACE
extern "C" int vn_debug_p2d_trace(long long *out /* host, 64 x 8 */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2d_trace), sizeof(long long) * 64 * 8);
}
#endif
