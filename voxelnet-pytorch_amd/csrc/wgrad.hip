// Weight gradient of the gather-GEMM convolution (autograd of model.py:134-153,
// 188-193: Conv2d/Conv3d/ConvTranspose2d .weight.grad):
//
//   dw[tap][n][k] += sum_m rows[m][n] * src[site(m, tap)][k]
//
// The reduction runs over the sites m, which is the SLOW dimension of both
// operands in memory (channels-last rows), so both MFMA operands need a
// transposed read.  gfx950 has one: ds_read_b64_tr_b16.  Slabs of 64 sites x
// {DN, DK} channels are staged row-major in LDS by LDS-DMA (per-lane gathered
// source rows; invalid rows read zeros through the buffer descriptor's bounds
// check), and each wave builds its v_mfma_f32_16x16x32_bf16 fragments with two
// transposed reads per operand tile.
// Grid: (row chunk, tap, (n,k) tile).  Every workgroup owns a DN x DK fp32 tile
// of one tap's dw, accumulated over its chunk of sites in registers.  Chunk
// partials go to a workspace slab with plain stores and k_wgrad_reduce adds them
// in a fixed order (no atomics: bit-reproducible, and fp32 atomics run at
// ~1.3 TB/s chip-wide against ~6 TB/s for stores).
// A 64-entry row table (site -> source/row byte offsets) is maintained
// incrementally by wave 0 one stage ahead, so the loaders do no index math.
#include "common.h"

namespace {

constexpr uint32_t WG_OOB = 0xFFFFF000u;
constexpr uint32_t WG_MAX_WINDOW = 0xFFFFE000u;

struct WGParams {
    const char *src;
    const char *rows;
    float *dw;
    int64_t sB, sD, sH, sW;   // src strides, elements
    int64_t rB, rD, rH, rW;   // rows strides, elements
    int32_t B, Ds, Hs, Ws, Dr, Hr, Wr;
    int32_t mulD, mulH, mulW, tmulD, tmulH, tmulW, padD, padH, padW;
    int32_t kD, kH, kW;
    int32_t C, N;             // real source / row channels
    int32_t x3;               // VN_F32X3: fp32 tiles, products as three bf16 MFMAs
    int32_t rows_per_chunk;   // multiple of 64
    int32_t tiles_k;          // number of DK tiles
    uint32_t src_bytes, rows_bytes;
    // row-list mode: the reduction rows are an explicit list of (b,d,h,w) coordinates; the rows tensor is then
    // a plain [n_rows][...] matrix (row index * rW) and the gathered site may be a strided TRANSPOSED one
    const int64_t *row_list;
    int64_t n_rows;
    const int32_t *row_count;   // device: number of valid list rows (<= n_rows = the list's capacity), or NULL
    int32_t divD, divH, divW;
    // partial sums: chunk c (blockIdx.x) stores its tiles at part + c * part_stride; NULL = a single chunk, which
    // adds into dw itself (one writer per element: no atomics either way)
    float *part;
    int64_t part_stride;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char *lds_wave_base, uint32_t voffset,
                                          uint32_t soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// swizzle of the 16-B chunk index inside a row, as a function of the row, so that the
// transposed 8-B reads of a 32-lane half hit 32 distinct bank pairs.
template <int ROW_BYTES, bool F32, bool PS = false>
__device__ __forceinline__ int chunk_swz(int row) {
    // split fp32 tiles (VN_F32X3S, rows >= 256 B): a 16-column MFMA tile is the hi granules c and c + 2 (lo: c + 1, c + 3) of
    // a row — BOTH even — so the eight rows {0..3, 8..11} of a 32-lane half must spread them with bits 0, 2 and 3
    if (PS) return (row & 1) | (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 3);
    if (F32) return (row & 3) << 2;   // fp32 tiles (rows >= 256 B) are read with ds_read_b32: 64-B shifts by row&3
    if (ROW_BYTES == 128) return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
    return ((row & 3) << 1) | (((row >> 3) & 1) << 3);   // 256-B rows
}

// wave tile in units of 16 channels; 2 x WKW waves (WKW = 4: eight waves on the same 128 x 128 tile, two waves per SIMD
// at one workgroup per CU — these launches have ~256 workgroups); taps per workgroup
// PS (with F32): both operands are stored split (VN_F32X3S: per 8 channels their eight hi bf16 parts, then their eight lo
// parts) — 32 sites per stage as for fp32, the fragments by transposed reads of the hi and lo granules as in the bf16 kernel,
// three MFMAs per tile pair and no split work (round 5; the in-register form below reads eight floats and spends 24 VALU
// operations per fragment)
template <int TN, int TK, bool F32, int TPB, int WKW = 2, bool PS = false>
__global__ void __launch_bounds__(128 * WKW, 2) k_wgrad(const WGParams p) {
    static_assert(!PS || F32, "split storage is fp32-sized");
    constexpr int NWV = 2 * WKW;                            // waves per workgroup
    constexpr int ESZ = F32 ? 4 : 2;
    constexpr int ROWS = F32 ? 32 : 64;                     // sites per stage (same bytes either way)
    constexpr int DN = 32 * TN, DK = 16 * WKW * TK;
    constexpr int RBN = DN * ESZ, RBK = DK * ESZ;           // LDS row bytes
    constexpr int TILE_N = ROWS * RBN, TILE_K = ROWS * RBK; // bytes per stage
    constexpr int STAGE = TILE_N + TPB * TILE_K;            // one `rows` slab shared by TPB gathered slabs
    constexpr int IN = TILE_N / (1024 * NWV), IK = TILE_K / (1024 * NWV);   // DMA instructions per wave per stage (per tile)
    static_assert(TILE_N % (1024 * NWV) == 0 && TILE_K % (1024 * NWV) == 0, "tile bytes per stage must split over the waves");
    constexpr int LPR_N = RBN / 16, LPR_K = RBK / 16;       // lanes per row
    constexpr int TW = TPB + 1;                             // table words per row: TPB source offsets + row offset
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *tbl = reinterpret_cast<uint32_t *>(smem + 2 * STAGE);   // [2][64][TW]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave / WKW, wk = wave % WKW;
    const int bx = blockIdx.x, by = blockIdx.y, tile = blockIdx.z, gx = gridDim.x;
    const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
    const int n0 = tn * DN, k0 = tk * DK;
    const int tap0 = by * TPB;          // TPB == 3: the three kW taps of one (kd, kh)
    const int td = tap0 / (p.kH * p.kW), th = (tap0 / p.kW) % p.kH, tw0 = tap0 % p.kW;
    const bool list = p.row_list != nullptr;
    int64_t M = list ? p.n_rows : (int64_t)p.B * p.Dr * p.Hr * p.Wr;
    if (list && p.row_count) {   // capacity launch: the chunks past the device-side count store all-zero partials
        const int64_t cnt = (int64_t)p.row_count[0];
        M = cnt < 0 ? 0 : (cnt < M ? cnt : M);
    }
    // counted list: the slabs of ROWS rows are dealt round-robin to the chunks (the launch is sized for the list's capacity;
    // contiguous ranges would leave the chunks past the count without work)
    const bool dealt = list && p.row_count;
    int64_t rbeg = (int64_t)bx * p.rows_per_chunk;
    if (rbeg >= M && !dealt) return;   // (the host sizes the grid so that every chunk has rows)
    int64_t rend = rbeg + p.rows_per_chunk;
    if (rend > M) rend = M;
    if (rend < rbeg) rend = rbeg;
    int nsteps = (int)((rend - rbeg + ROWS - 1) / ROWS);
    int64_t slab_stride = ROWS;        // rows between two consecutive slabs of this chunk
    if (dealt) {
        const int64_t nslabs = (M + ROWS - 1) / ROWS;
        nsteps = (int)(nslabs > bx ? (nslabs - bx + gx - 1) / gx : 0);
        rbeg = (int64_t)bx * ROWS;
        rend = M;
        slab_stride = (int64_t)gx * ROWS;
    }
    const int nstages = nsteps;

    const __amdgpu_buffer_rsrc_t rs_s = vn_uniform_rsrc(p.src, p.src_bytes);
    const __amdgpu_buffer_rsrc_t rs_r = vn_uniform_rsrc(p.rows, p.rows_bytes);

    // ---- row table producer state (wave 0: lane <-> row of the current slab) ----
    int cb = 0, cd = 0, ch = 0, cw = 0;
    int cm32 = lane;            // the site this lane tracks, relative to the chunk
    const int rows_in_chunk = (int)(rend - rbeg);
    int tstep = 0;              // slab index its coordinates stand for
    const uint32_t rBb = (uint32_t)(p.rB * ESZ), rDb = (uint32_t)(p.rD * ESZ), rHb = (uint32_t)(p.rH * ESZ), rWb = (uint32_t)(p.rW * ESZ);
    const uint32_t sBb = (uint32_t)(p.sB * ESZ), sDb = (uint32_t)(p.sD * ESZ), sHb = (uint32_t)(p.sH * ESZ), sWb = (uint32_t)(p.sW * ESZ);
    if (wave == 0 && !list) {
        int64_t t = rbeg + lane;
        cw = (int)(t % p.Wr); t /= p.Wr;
        ch = (int)(t % p.Hr); t /= p.Hr;
        cd = (int)(t % p.Dr);
        cb = (int)(t / p.Dr);
    }
    auto table_write = [&](int stage_idx) {
        // wave 0 only: entries for slab stage_idx into tbl[stage_idx & 1]
        const int step = stage_idx;
        uint32_t so[TPB], ro = WG_OOB;
#pragma unroll
        for (int j = 0; j < TPB; ++j) so[j] = WG_OOB;
        if (list) {
            const int64_t m = rbeg + (int64_t)step * slab_stride + lane;
            if (m < rend && lane < ROWS) {
                const int64_t *rc = p.row_list + m * 4;
                const int b = (int)rc[0];
                int sd = (int)rc[1] * p.mulD + td * p.tmulD - p.padD;
                int sh = (int)rc[2] * p.mulH + th * p.tmulH - p.padH;
                ro = (uint32_t)(m * p.rW * ESZ);
                const bool dh_ok = sd >= 0 && sh >= 0 && sd % p.divD == 0 && sh % p.divH == 0;
                sd /= p.divD; sh /= p.divH;
#pragma unroll
                for (int j = 0; j < TPB; ++j) {
                    int sw = (int)rc[3] * p.mulW + (tw0 + j) * p.tmulW - p.padW;
                    const bool ok = dh_ok && sw >= 0 && sw % p.divW == 0;
                    sw /= p.divW;
                    if (ok && sd < p.Ds && sh < p.Hs && sw < p.Ws)
                        so[j] = (uint32_t)(((int64_t)b * p.sB + (int64_t)sd * p.sD + (int64_t)sh * p.sH + (int64_t)sw * p.sW) * ESZ);
                }
            }
        } else {
            // 32-bit, branch-light arithmetic (wave 0 does this on the critical path of every stage: measured 870 clk
            // with 64-bit products and loops).  Byte offsets fit 32 bits: the buffer windows are < 4 GB.
            while (tstep < step) {   // advance by ROWS sites (normally exactly one slab)
                cm32 += ROWS;
                cw += ROWS;
#pragma unroll
                for (int k = 0; k < 3; ++k) {            // ROWS / Wr <= 3 wraps (Wr >= 22)
                    const bool wrap = cw >= p.Wr;
                    cw -= wrap ? p.Wr : 0;
                    ch += wrap ? 1 : 0;
                }
                while (cw >= p.Wr) { cw -= p.Wr; ++ch; }   // (narrower rows)
                while (ch >= p.Hr) { ch -= p.Hr; ++cd; }
                while (cd >= p.Dr) { cd -= p.Dr; ++cb; }
                ++tstep;
            }
            if (cm32 < rows_in_chunk) {
                const int sd = cd * p.mulD + td * p.tmulD - p.padD;
                const int sh = ch * p.mulH + th * p.tmulH - p.padH;
                ro = (uint32_t)cb * rBb + (uint32_t)cd * rDb + (uint32_t)ch * rHb + (uint32_t)cw * rWb;
                if ((unsigned)sd < (unsigned)p.Ds && (unsigned)sh < (unsigned)p.Hs) {
                    const uint32_t base = (uint32_t)cb * sBb + (uint32_t)sd * sDb + (uint32_t)sh * sHb;
#pragma unroll
                    for (int j = 0; j < TPB; ++j) {
                        const int sw = cw * p.mulW + (tw0 + j) * p.tmulW - p.padW;
                        if ((unsigned)sw < (unsigned)p.Ws) so[j] = base + (uint32_t)sw * sWb;
                    }
                }
            }
        }
        uint32_t *e = tbl + ((stage_idx & 1) * 64 + lane) * TW;
#pragma unroll
        for (int j = 0; j < TPB; ++j) e[j] = so[j];
        e[TPB] = ro;
    };

    // per-lane DMA geometry: instruction i of this wave covers LDS bytes ((i*4+wave)*1024 .. +1023) of a tile
    auto stage = [&](int sidx, int buf) {
        const uint32_t s_col = __builtin_amdgcn_readfirstlane((uint32_t)(k0 * ESZ));
        const uint32_t r_col = __builtin_amdgcn_readfirstlane((uint32_t)(n0 * ESZ));
        const uint32_t *t = tbl + (sidx & 1) * 64 * TW;
        char *ln = smem + buf * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < IN; ++i) {
            const int r = ((i * NWV + wave) * 1024) / RBN + lane / LPR_N;
            const int c = (lane % LPR_N) ^ chunk_swz<RBN, F32, PS>(r);
            const uint32_t ro = t[r * TW + TPB];
            const bool ok = ro != WG_OOB && (n0 + c * (16 / ESZ)) < p.N;
            lds_dma16(rs_r, ln + i * (1024 * NWV), ok ? ro + (uint32_t)c * 16u : WG_OOB, r_col);
        }
#pragma unroll
        for (int j = 0; j < TPB; ++j) {
            char *lk = smem + buf * STAGE + TILE_N + j * TILE_K + wave * 1024;
#pragma unroll
            for (int i = 0; i < IK; ++i) {
                const int r = ((i * NWV + wave) * 1024) / RBK + lane / LPR_K;
                const int c = (lane % LPR_K) ^ chunk_swz<RBK, F32, PS>(r);
                const uint32_t so = t[r * TW + j];
                const bool ok = so != WG_OOB && (k0 + c * (16 / ESZ)) < p.C;
                lds_dma16(rs_s, lk + i * (1024 * NWV), ok ? so + (uint32_t)c * 16u : WG_OOB, s_col);
            }
        }
    };

    f32x4_t acc[TPB][TN][TK];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane geometry: lane = 16g + 4q + pp -> row (8g + q [+4]), 8-B piece pp of a 16-column tile
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;

    if (wave == 0 && nstages > 0) table_write(0);
    __syncthreads();
    if (nstages > 0) stage(0, 0);
    if (wave == 0 && nstages > 1) table_write(1);
    for (int s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nstages) stage(s + 1, buf ^ 1);
        if (wave == 0 && s + 2 < nstages) table_write(s + 2);
        const char *ln = smem + buf * STAGE;
        const char *lk0 = smem + buf * STAGE + TILE_N;
        if constexpr (PS) {
            // split operands: lane = 16g + 4q + pp reads, of a 16-channel tile (64 B of a row), the 8-B piece pp of rows
            // 8g + q and 8g + q + 4: channels 4 pp .. 4 pp + 3 — hi at byte (pp >> 1) * 32 + (pp & 1) * 8, lo 16 B further
            typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
            static_assert(ROWS == 32, "one 32-site MFMA per stage");
            const int r0 = g * 8 + q, r1 = r0 + 4;
            const int sw0 = chunk_swz<RBN, true, true>(r0), sw1 = chunk_swz<RBN, true, true>(r1);
            const int pc = (pp >> 1) * 2, ph = (pp & 1) * 8;          // 16-B chunk inside the tile, 8-B half
            auto frag = [&](const char *tile, int row_bytes, int t16, int lo) {
                const int c16 = t16 * 4 + pc + lo;
                const char *p0 = tile + r0 * row_bytes + ((c16 ^ sw0) << 4) + ph;
                const char *p1 = tile + r1 * row_bytes + ((c16 ^ sw1) << 4) + ph;
                const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p0);
                const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p1);
                const s16x8_t t8 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                return __builtin_bit_cast(bf16x8_t, t8);
            };
            bf16x8_t ah[TN], al[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                ah[i] = frag(ln, RBN, wn * TN + i, 0);
                al[i] = frag(ln, RBN, wn * TN + i, 1);
            }
#pragma unroll
            for (int t = 0; t < TPB; ++t)
#pragma unroll
                for (int j = 0; j < TK; ++j) {
                    const bf16x8_t bh = frag(lk0 + t * TILE_K, RBK, wk * TK + j, 0);
                    const bf16x8_t bl = frag(lk0 + t * TILE_K, RBK, wk * TK + j, 1);
#pragma unroll
                    for (int i = 0; i < TN; ++i) acc[t][i][j] = vn_mfma_x3(ah[i], al[i], bh, bl, acc[t][i][j]);
                }
        } else if constexpr (F32) {
            // v_mfma_f32_16x16x4_f32: lane (c = lane&15, kq = lane>>4) supplies element [site 4s+kq][col c]
            const int fc = lane & 15, kq = lane >> 4;
            if (p.x3) {
                // fp32x3: the lane's eight sites 4 ss + kq (ss = 0..7) of a column are its eight k values of ONE
                // v_mfma_f32_16x16x32_bf16 (same (ss, kq) <-> k assignment for both operands), split hi / lo in registers
                static_assert(ROWS == 32, "eight sites per lane");
                bf16x8_t ah[TN], al[TN];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int col = (wn * TN + i) * 16 + fc;
                    float v[8];
#pragma unroll
                    for (int ss = 0; ss < 8; ++ss) {
                        const int r = ss * 4 + kq;
                        v[ss] = *reinterpret_cast<const float *>(ln + r * RBN + ((((col >> 2) ^ chunk_swz<RBN, true>(r))) << 4) + (col & 3) * 4);
                    }
                    vn_split8(v, ah[i], al[i]);
                }
#pragma unroll
                for (int t = 0; t < TPB; ++t)
#pragma unroll
                    for (int j = 0; j < TK; ++j) {
                        const int col = (wk * TK + j) * 16 + fc;
                        float v[8];
#pragma unroll
                        for (int ss = 0; ss < 8; ++ss) {
                            const int r = ss * 4 + kq;
                            v[ss] = *reinterpret_cast<const float *>(lk0 + t * TILE_K + r * RBK + ((((col >> 2) ^ chunk_swz<RBK, true>(r))) << 4) + (col & 3) * 4);
                        }
                        bf16x8_t bh, bl;
                        vn_split8(v, bh, bl);
#pragma unroll
                        for (int i = 0; i < TN; ++i) acc[t][i][j] = vn_mfma_x3(ah[i], al[i], bh, bl, acc[t][i][j]);
                    }
            } else
#pragma unroll
            for (int ss = 0; ss < ROWS / 4; ++ss) {
                const int r = ss * 4 + kq;
                const int sw = chunk_swz<RBN, true>(r);     // same function for both tiles (row & 3)
                float a[TN];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int col = (wn * TN + i) * 16 + fc;
                    a[i] = *reinterpret_cast<const float *>(ln + r * RBN + ((((col >> 2) ^ sw)) << 4) + (col & 3) * 4);
                }
#pragma unroll
                for (int t = 0; t < TPB; ++t) {
                    float b[TK];
#pragma unroll
                    for (int j = 0; j < TK; ++j) {
                        const int col = (wk * TK + j) * 16 + fc;
                        b[j] = *reinterpret_cast<const float *>(lk0 + t * TILE_K + r * RBK + ((((col >> 2) ^ sw)) << 4) + (col & 3) * 4);
                    }
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TK; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[t][i][j], 0, 0, 0);
                }
            }
        } else {
            typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int r0 = ks * 32 + g * 8 + q, r1 = r0 + 4;
                bf16x8_t a[TN];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int col = (wn * TN + i) * 16 + pp * 4;          // element column inside the tile
                    const int c16 = col >> 3, half = (col >> 2) & 1;      // 16-B chunk, 8-B half
                    const char *p0 = ln + r0 * RBN + ((c16 ^ chunk_swz<RBN, false>(r0)) << 4) + half * 8;
                    const char *p1 = ln + r1 * RBN + ((c16 ^ chunk_swz<RBN, false>(r1)) << 4) + half * 8;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p0);
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p1);
                    const s16x8_t t8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    a[i] = __builtin_bit_cast(bf16x8_t, t8);
                }
#pragma unroll
                for (int t = 0; t < TPB; ++t) {
                    bf16x8_t b[TK];
#pragma unroll
                    for (int j = 0; j < TK; ++j) {
                        const int col = (wk * TK + j) * 16 + pp * 4;
                        const int c16 = col >> 3, half = (col >> 2) & 1;
                        const char *p0 = lk0 + t * TILE_K + r0 * RBK + ((c16 ^ chunk_swz<RBK, false>(r0)) << 4) + half * 8;
                        const char *p1 = lk0 + t * TILE_K + r1 * RBK + ((c16 ^ chunk_swz<RBK, false>(r1)) << 4) + half * 8;
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p0);
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p1);
                        const s16x8_t t8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        b[j] = __builtin_bit_cast(bf16x8_t, t8);
                    }
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TK; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[t][i][j], 0, 0, 0);
                }
            }
        }
    }

    // D[n][k]: n = (lane>>4)*4 + e, k = lane&15
    float *dst = p.part ? p.part + (int64_t)bx * p.part_stride : p.dw;
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) {
                const int k = k0 + (wk * TK + j) * 16 + (lane & 15);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + (wn * TN + i) * 16 + (lane >> 4) * 4 + e;
                    if (n < p.N && k < p.C) {
                        float *q = dst + ((int64_t)(tap0 + t) * p.N + n) * p.C + k;
                        *q = p.part ? acc[t][i][j][e] : *q + acc[t][i][j][e];
                    }
                }
            }
}

// dw[i] += sum over chunks (fixed order) of part[c][i]
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float *__restrict__ part, int chunks, int64_t n4,
                                                      float *__restrict__ dw) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < chunks; ++c) {
            const float4 v = reinterpret_cast<const float4 *>(part)[(int64_t)c * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4 d = reinterpret_cast<float4 *>(dw)[i];
        d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
        reinterpret_cast<float4 *>(dw)[i] = d;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of 3x3 (x kD), stride-1-in-H/W convolutions with the shifted source rows read from an LDS halo
// patch ("patch" form, cf. k_conv_patch): a stage is a 4 x 16 rectangle of sites of one (batch, depth) plane — its
// 64 `rows` vectors and the 6 x 18 source patch around it — and all nine (kh,kw) taps of one kd are accumulated
// from it.  DMA pieces per stage: (64 + 108)/32 per wave for 72 MFMAs per wave, against 8 pieces for 24 MFMAs in
// the three-tap row form above, whose step time is the DMA issue time.  Zero padding = patch rows that were out of
// range (LDS-DMA wrote zeros).  bf16, 64 x 64 channel tiles, 2 x 2 waves of 32 x 32.
struct WPParams {
    const char *src, *rows;
    float *part;                 // chunk partials: part + chunk * part_stride
    int64_t part_stride;
    int64_t sB, sD, sH, sW, rB, rD, rH, rW;   // elements
    int32_t B, Ds, Hs, Ws, Dr, Hr, Wr;
    int32_t mulD, padD, kD;      // source plane = d * mulD + kd - padD
    int32_t C, N, tiles_k;
    int32_t tiles_x, tiles_y;    // 4 x 16 tiles per plane
    int32_t tiles_per_chunk;
    uint32_t src_bytes, rows_bytes;
    // where the 16-B chunk of channels 8c .. 8c+7 of a site lives: byte c * cb + part.  Plain bf16 rows: cb = 16, part = 0.
    // One bf16 HALF of a tensor in split fp32 storage (VN_F32X3S: per 8 channels 16 B of hi parts, then 16 B of lo parts):
    // cb = 32, part = 0 (hi) / 16 (lo) — the three bf16 passes of an fp32x3 weight gradient read the operands where the
    // BatchNorm passes stored them (round 5; round 4 made [hi | lo] row copies first)
    uint32_t s_cb, s_part, r_cb, r_part;
};

// NH = number of 64-channel halves of the `rows` tile: NH = 1 is the 64 x 64 x nine-tap tile of the 64-channel Conv3d
// layers (four waves) — the only instantiation the library ships since round 5; NH = 2 (round 4, measured slower in the step
// and retired: wgrad_patch_enabled) a 128 x 64 x nine-tap tile on EIGHT waves for the 128- / 256-channel 2-D layers:
// wave (wn, wk) owns rows channels 64 wn .. 64 wn + 63 x source channels 16 wk .. 16 wk + 15 of all nine taps (144
// accumulator registers, two waves per SIMD).  A stage of 64 sites stages 16 KB of `rows` + 13.8 KB of patch for
// 9 x 128 x 64 x 64 MACs: 6.3 staged bytes per kMAC against 30.5 for the single-tap 128 x 128 row form (k_wgrad<4,2,.,1,4>),
// whose step time is its LDS-DMA issue + per-stage synchronisation.  The partial-sum traffic of a weight gradient is
// 4 B x (tile elements) per workgroup whatever the tile shape, i.e. it only depends on the sites per workgroup; at equal
// sites per workgroup this tile needs 4.5x fewer workgroups than the single-tap one (2 instead of 9 per row chunk at
// 128 -> 128 channels): the launch leaves three quarters of the CUs to the main stream's data gradients.
template <int NH>
__global__ void __launch_bounds__(256 * NH, 2) k_wgrad_patch(const WPParams p) {
    // LDS patch: 6 lines of 32 rows (18 used: 16 sites + halo) x 128 B.  The line pitch of 32 rows keeps the bank
    // swizzle (a function of row bits 1 and 3) independent of the line, so a tap shift (th lines, tw rows) changes a
    // lane's read address by a compile-time immediate (th) plus one of three precomputed per-lane offsets (tw): the
    // 72 transposed reads of a stage need no address arithmetic.
    constexpr int NWV = 4 * NH;                              // waves per workgroup
    constexpr int TH = 4, TW = 16, PH = TH + 2, LP = 32;      // 64 sites
    constexpr int RB = 128;                                  // patch row bytes: 64 bf16 source channels
    constexpr int RBN = 128 * NH;                            // `rows` tile row bytes: 64 NH bf16 channels
    constexpr int TILE_N = 64 * RBN;                         // 8 / 16 KiB
    constexpr int PATCH = PH * LP * RB;                      // 24 KiB (6 x 18 rows of it loaded)
    constexpr int STAGE = TILE_N + PATCH;
    constexpr int IN = TILE_N / (1024 * NWV);                // DMA instructions per wave: rows tile (2)
    constexpr int RPI = 1024 / RBN, LPR = RBN / 16;          // rows per DMA instruction, lanes per row
    constexpr int NPP = PH * 3;                              // patch pieces: 3 per line (rows 0-7, 8-15, 16-23)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave (wn, wk) owns 64 `rows` channels x source channels 16wk..16wk+15 of every tap: the four A (rows) fragments are
    // read once per K half and shared by the nine taps, a tap costs ONE B fragment (2 transposed reads) per 4 MFMAs —
    // the loop is LDS-read-bound (2 x 2 waves of 32 x 32 needed 160 KB of reads per stage and workgroup, this 104 KB)
    const int wn = wave >> 2, wk = wave & 3;
    const int tile = blockIdx.z;
    const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
    const int n0 = tn * 64 * NH, k0 = tk * 64;
    const int kd = blockIdx.y;
    const int64_t ntiles = (int64_t)p.B * p.Dr * p.tiles_y * p.tiles_x;
    const int64_t tbeg = (int64_t)blockIdx.x * p.tiles_per_chunk;
    int64_t tend = tbeg + p.tiles_per_chunk;
    if (tend > ntiles) tend = ntiles;
    const int nst = (int)(tend - tbeg);                      // >= 1 by the grid size

    const __amdgpu_buffer_rsrc_t rs_s = vn_uniform_rsrc(p.src, p.src_bytes);
    const __amdgpu_buffer_rsrc_t rs_r = vn_uniform_rsrc(p.rows, p.rows_bytes);
    const uint32_t s_col = (uint32_t)(k0 >> 3) * p.s_cb + p.s_part, r_col = (uint32_t)(n0 >> 3) * p.r_cb + p.r_part;

    // ---- loader state.  Per lane and piece, constant over the stages: the site / patch cell it loads (relative to the
    // tile origin), its byte offset relative to the tile's first site, its 16-B chunk.  Per stage only the tile origin
    // changes (kept incrementally: no divisions in the loop), so a piece costs a few compares and one add.
    int r_py[IN], r_px[IN];
    uint32_t r_dlt[IN], r_chunk[IN];
#pragma unroll
    for (int i = 0; i < IN; ++i) {
        const int r = (i * NWV + wave) * RPI + lane / LPR;                   // site 0..63
        const int c = (lane % LPR) ^ chunk_swz<RBN, false>(r);
        r_py[i] = r >> 4; r_px[i] = r & 15;
        r_dlt[i] = (uint32_t)((((int64_t)(r >> 4)) * p.rH + (int64_t)(r & 15) * p.rW) * 2) + (uint32_t)c * p.r_cb;
        r_chunk[i] = (n0 + c * 8) < p.N ? 1u : 0u;
    }
    constexpr int IPP = (NPP + NWV - 1) / NWV;
    int s_qy[IPP], s_qx[IPP];
    int32_t s_dlt[IPP];
    uint32_t s_ok[IPP], s_lds[IPP];
#pragma unroll
    for (int i = 0; i < IPP; ++i) {
        const int piece = i * NWV + wave;
        const int ql = piece / 3, cg = piece - ql * 3;
        const int qx = cg * 8 + lane / 8;                                    // column inside the line (0..23; 18.. unused)
        const int prow = ql * LP + qx;
        const int c = (lane % 8) ^ chunk_swz<RB, false>(prow);
        s_qy[i] = ql - 1; s_qx[i] = qx - 1;                                  // source cell relative to the tile origin
        s_dlt[i] = (int32_t)((((int64_t)(ql - 1)) * p.sH + (int64_t)(qx - 1) * p.sW) * 2) + c * (int32_t)p.s_cb;
        s_ok[i] = (piece < NPP && qx < TW + 2 && (k0 + c * 8) < p.C) ? 1u : 0u;
        s_lds[i] = (uint32_t)((ql * LP + cg * 8) * RB);
    }
    // tile coordinates of the NEXT stage to be issued (tile index tbeg + issued)
    int tx, ty, td, tb;
    {
        int64_t t = tbeg;
        tx = (int)(t % p.tiles_x); t /= p.tiles_x;
        ty = (int)(t % p.tiles_y); t /= p.tiles_y;
        td = (int)(t % p.Dr);
        tb = (int)(t / p.Dr);
    }
    auto stage = [&](int buf) {
        const int y0 = ty * TH, x0 = tx * TW;
        const int sd = td * p.mulD + kd - p.padD;
        const bool plane_ok = (unsigned)sd < (unsigned)p.Ds;  // block-uniform
        const uint32_t rbase = (uint32_t)(((int64_t)tb * p.rB + (int64_t)td * p.rD + (int64_t)y0 * p.rH + (int64_t)x0 * p.rW) * 2);
        const uint32_t sbase = (uint32_t)(((int64_t)tb * p.sB + (int64_t)sd * p.sD + (int64_t)y0 * p.sH + (int64_t)x0 * p.sW) * 2);
        char *ln = smem + buf * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < IN; ++i) {
            const bool ok = plane_ok && r_chunk[i] && y0 + r_py[i] < p.Hr && x0 + r_px[i] < p.Wr;
            lds_dma16(rs_r, ln + i * (1024 * NWV), ok ? rbase + r_dlt[i] : WG_OOB, r_col);
        }
        char *lp = smem + buf * STAGE + TILE_N;
#pragma unroll
        for (int i = 0; i < IPP; ++i) {
            if (i * NWV + wave >= NPP) break;
            const bool ok = plane_ok && s_ok[i] && (unsigned)(y0 + s_qy[i]) < (unsigned)p.Hs &&
                            (unsigned)(x0 + s_qx[i]) < (unsigned)p.Ws;
            lds_dma16(rs_s, lp + s_lds[i], ok ? sbase + (uint32_t)s_dlt[i] : WG_OOB, s_col);
        }
        // advance to the next tile
        if (++tx == p.tiles_x) {
            tx = 0;
            if (++ty == p.tiles_y) {
                ty = 0;
                if (++td == p.Dr) { td = 0; ++tb; }
            }
        }
    };

    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane geometry: lane = 16g + 4q + pp -> site (ks*32 + 8g + q [+4]), 8-B piece pp of a 16-column tile
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
    // rows tile: sites r0 = ks*32 + 8g + q and r0 + 4 (ks*32 rows = ks*32*RBN B: an immediate)
    int aoff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = (wn * 4 + i) * 16 + pp * 4, c16 = col >> 3, half = (col >> 2) & 1;
        const int r0 = g * 8 + q;
        aoff[i][0] = r0 * RBN + ((c16 ^ chunk_swz<RBN, false>(r0)) << 4) + half * 8;
        aoff[i][1] = (r0 + 4) * RBN + ((c16 ^ chunk_swz<RBN, false>(r0 + 4)) << 4) + half * 8;
    }
    // patch: site (line ks*2 + (g>>1), column (g&1)*8 + q [+4]) shifted by tap (th, tw) -> patch row (line + th)*32 + col + tw
    int boff[3][2];
#pragma unroll
    for (int tw = 0; tw < 3; ++tw)
#pragma unroll
        for (int wq = 0; wq < 2; ++wq) {
            const int colw = (g & 1) * 8 + q + tw + 4 * wq;
            const int sw = chunk_swz<RB, false>(colw);
            const int col = wk * 16 + pp * 4, c16 = col >> 3, half = (col >> 2) & 1;
            boff[tw][wq] = ((g >> 1) * LP + colw) * RB + ((c16 ^ sw) << 4) + half * 8;
        }

    stage(0);
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nst) stage(buf ^ 1);
        const char *ln = smem + buf * STAGE;
        const char *lp = ln + TILE_N;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(ln + ks * 32 * RBN + aoff[i][0]));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(ln + ks * 32 * RBN + aoff[i][1]));
                a[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int th = t / 3, tw = t % 3;
                const char *lt = lp + (ks * 2 + th) * LP * RB;               // compile-time offset
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(lt + boff[tw][0]));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(lt + boff[tw][1]));
                const bf16x8_t bq = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                // operands swapped (source-channel fragment first): the accumulator is the TRANSPOSED tile D[k][n], so a lane
                // holds four consecutive source channels k of one rows channel n — the epilogue stores 16 B per lane
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, a[i], acc[t][i], 0, 0, 0);
            }
        }
    }
    // D[k][n]: k = (lane>>4)*4 + e, n = lane&15 ; tap index kd*9 + t.  One float4 per accumulator (36 store instructions per
    // wave instead of 144 dword ones: the store tail of a workgroup's 147 / 295 KB is issue-bound, cdna_hip_programming.md T21)
    float *dst = p.part + (int64_t)blockIdx.x * p.part_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + wk * 16 + (lane >> 4) * 4;
            const int n = n0 + (wn * 4 + i) * 16 + (lane & 15);
            if (n < p.N && k < p.C)     // (C is a multiple of 8: k .. k + 3 are in range together)
                *reinterpret_cast<f32x4_t *>(dst + ((int64_t)(kd * 9 + t) * p.N + n) * p.C + k) = acc[t][i];
        }
}

template <int TN, int TK, bool F32, int TPB, int WKW = 2, bool PS = false>
int launch_wgrad(const WGParams &p, dim3 grid, hipStream_t st) {
    constexpr int DN = 32 * TN, DK = 16 * WKW * TK;
    // same stage bytes for bf16 (64 sites) and fp32 (32 sites); + the row table
    constexpr size_t lds = 2u * 64u * (DN + TPB * DK) * 2u + 2u * 64u * (TPB + 1) * 4u;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad<TN, TK, F32, TPB, WKW, PS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    k_wgrad<TN, TK, F32, TPB, WKW, PS><<<grid, 128 * WKW, lds, st>>>(p);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

}  // namespace

static int wgrad_impl(const void *src, const void *rows, float *dw_packed, const vnConv *g, int32_t split,
                      const int64_t *row_list, int64_t n_rows, void *workspace, size_t workspace_bytes, vnStream stream,
                      int32_t *chunks_out = nullptr, const int32_t *row_count = nullptr, int split_pass = -1);

// tiling of one weight-gradient problem (shared by the launcher and the workspace query)
struct WGPlan {
    bool tri, n128, k128;
    int tiles_n, tiles_k, groups;
    int64_t chunks;       // preferred number of row chunks (partial-sum slabs)
    int64_t dw_elems;
};
static WGPlan wgrad_plan(const vnConv *g, int32_t split, int64_t M) {
    WGPlan w{};
    const bool f32 = g->dtype != VN_BF16;   // (VN_F32X3: the fp32 kernels)
    const int taps = g->kD * g->kH * g->kW;
    // three-tap mode (bf16): one workgroup owns the three kW taps of a (kd,kh) pair, a 64-row `rows` slab is
    // staged and fragment-read once for all three -> 3x the MFMA work per barrier
    w.tri = !f32 && !split && g->kW == 3 && g->Cr <= 64;   // (wider outputs: the 128x128 single-tap tile wins)
    w.n128 = !w.tri && g->Cr > 64;
    w.k128 = g->Cs > 64;
    w.tiles_n = (int)vn_ceil_div(g->Cr, w.n128 ? 128 : 64);
    w.tiles_k = (int)vn_ceil_div(g->Cs, w.k128 ? 128 : 64);
    w.groups = w.tri ? taps / 3 : taps;
    w.dw_elems = (int64_t)taps * g->Cr * g->Cs;
    // ~256 workgroups (one per CU; the data-gradient launches of the main stream run beside them) and at least 10
    // slabs of 64 sites per chunk: more chunks only add partial-tile traffic (each chunk stores DN x DK x taps fp32
    // and the batched unpack reads it back), fewer leave CUs idle.  Measured in the full step: 128 / 192 / 256 / 384 /
    // 512 / 768 workgroups -> 357 / 368 / 373 / 370 / 368 / 364 point-clouds/s.
    // fp32x3 (VALU / LDS-latency-bound stages, one workgroup does not fill a CU): 192 / 256 / 384 / 512 -> 246 / 259 / 265 /
    // 262 point-clouds/s in that mode (round 4)
    static const int knob = vn_knob("VN_WG_BLOCKS", 0);   // tuning aid; 0 = 256 (384 for fp32x3)
    const int target = knob > 0 ? knob : ((g->dtype == VN_F32X3 || g->dtype == VN_F32X3S) ? 384 : 256);
    int64_t chunks = target / ((int64_t)w.groups * w.tiles_n * w.tiles_k);
    const int64_t slabs = vn_ceil_div(M, 64);
    if (chunks > slabs / 10) chunks = slabs / 10;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    w.chunks = chunks;
    return w;
}

// ---- patch form (k_wgrad_patch): eligibility and chunking
struct WPPlan {
    bool ok;
    int nh;               // 64-channel halves of the `rows` tile (k_wgrad_patch<NH>)
    int tiles_x, tiles_y, tiles_n, tiles_k;
    int64_t ntiles, chunks;
};
static int wgrad_patch_enabled() {   // tuning aid: VN_WGRAD_PATCH=0 keeps the row form everywhere, 1 (default) = the patch form on
    // the 64-channel Conv3d layers.  (Round 4's 128 x 64 nine-tap tile for the wider 2-D layers — 626 / 802 TFLOP/s alone, 526-554
    // against 559 point-clouds/s in the step: 4.5x the partial-sum bytes per workgroup — left the library in round 5;
    // DESIGN_HISTORY.md keeps the numbers, git the code.)
    static const int v = vn_knob("VN_WGRAD_PATCH", 1);
    return v;
}
static WPPlan wgrad_patch_plan(const vnConv *g, int32_t split, bool list) {
    WPPlan w{};
    if (!wgrad_patch_enabled() || list || split || g->dtype != VN_BF16) return w;
    if (g->kH != 3 || g->kW != 3 || g->mulH != 1 || g->mulW != 1 || g->tmulH != 1 || g->tmulW != 1 || g->padH != 1 ||
        g->padW != 1 || g->tmulD != 1 || g->divD != 1 || g->divH != 1 || g->divW != 1)
        return w;
    if (g->Hs != g->Hr || g->Ws != g->Wr) return w;
    // measured (round 2): a win for the 64-channel Conv3d layers (1.3-1.4x) at 400 x 352
    if (g->Cr > 64 || g->Hr < 128 || g->Wr < 128 || g->Cs > 64) return w;
    w.nh = 1;
    w.tiles_x = (int)vn_ceil_div(g->Wr, 16);
    w.tiles_y = (int)vn_ceil_div(g->Hr, 4);
    w.tiles_n = (int)vn_ceil_div(g->Cr, 64 * w.nh);
    w.tiles_k = (int)vn_ceil_div(g->Cs, 64);
    w.ntiles = (int64_t)g->B * g->Dr * w.tiles_y * w.tiles_x;
    // (one workgroup per CU: in the step these launches run on the side stream beside the data gradients;
    //  128 ... 512 workgroups measure within 1 %, 768 is 3 % slower)
    static const int ptarget = vn_knob("VN_WGP_BLOCKS", 256);   // tuning aid
    int64_t chunks = ptarget / ((int64_t)g->kD * w.tiles_n * w.tiles_k);
    const int64_t min_stages = 8;
    if (chunks > w.ntiles / min_stages) chunks = w.ntiles / min_stages;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    w.chunks = chunks;
    w.ok = true;
    return w;
}

extern "C" size_t vn_conv_wgrad_workspace_bytes(const vnConv *g, int32_t split, int64_t n_rows) {
    if (!g || g->B <= 0 || g->Dr <= 0 || g->Hr <= 0 || g->Wr <= 0 || g->Cs <= 0 || g->Cr <= 0) return 0;
    if (g->kD < 1 || g->kH < 1 || g->kW < 1) return 0;
    const int64_t M = n_rows > 0 ? n_rows : (int64_t)g->B * g->Dr * g->Hr * g->Wr;
    const WGPlan w = wgrad_plan(g, split, M);
    int64_t chunks = w.chunks;
    const WPPlan wp = wgrad_patch_plan(g, split, n_rows > 0);
    if (wp.ok && wp.chunks > chunks) chunks = wp.chunks;
    return (size_t)chunks * (size_t)w.dw_elems * sizeof(float);   // (>= one chunk: vn_conv_wgrad_partials always uses it)
}

// Which kernel vn_conv_wgrad / vn_conv_wgrad_partials pick for a geometry: 200 = k_wgrad_patch<1>; else
// 1000 * (three-tap mode) + 10 * TN + TK of k_wgrad<TN, TK, ., .> (tile = 32 TN x 32 TK channels).
extern "C" int32_t vn_conv_wgrad_plan_id(const vnConv *g, int32_t split, int64_t n_rows) {
    if (!g || g->B <= 0 || g->Dr <= 0 || g->Hr <= 0 || g->Wr <= 0 || g->Cs <= 0 || g->Cr <= 0) return -1;
    const int64_t M = n_rows > 0 ? n_rows : (int64_t)g->B * g->Dr * g->Hr * g->Wr;
    const WPPlan wp = wgrad_patch_plan(g, split, n_rows > 0);
    if (wp.ok) return 200;
    const WGPlan w = wgrad_plan(g, split, M);
    if (w.tri) return 1000 + 10 * 2 + (w.k128 ? 4 : 2);
    return 10 * (w.n128 ? 4 : 2) + (w.k128 ? 4 : 2);
}

extern "C" int vn_conv_wgrad(const void *src, const void *rows, float *dw_packed, const vnConv *g, int32_t split,
                             void *workspace, size_t workspace_bytes, vnStream stream) {
    return wgrad_impl(src, rows, dw_packed, g, split, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int vn_conv_wgrad_partials(const void *src, const void *rows, const vnConv *g, int32_t split,
                                      const int64_t *row_list, int64_t n_rows, void *workspace, size_t workspace_bytes,
                                      int32_t *chunks, vnStream stream) {
    VN_CHECK_ARG(chunks && workspace && (row_list || n_rows == 0));
    if (row_list && n_rows == 0) {     // no rows: one all-zero partial
        const size_t bytes = (size_t)g->kD * g->kH * g->kW * g->Cr * g->Cs * sizeof(float);
        if (workspace_bytes < bytes) return VN_EWORKSPACE;
        VN_HIP(hipMemsetAsync(workspace, 0, bytes, vn_stream(stream)));
        *chunks = 1;
        return VN_OK;
    }
    return wgrad_impl(src, rows, static_cast<float *>(workspace), g, row_list ? 0 : split, row_list, n_rows, workspace,
                      workspace_bytes, stream, chunks);
}

// vn_conv_wgrad_partials over a row list whose length is only known on the device: row_cap = the list's capacity (sizes
// the launch and the chunking), *row_count (device) the valid rows; chunks past the count store all-zero partials.
extern "C" int vn_conv_wgrad_partials_counted(const void *src, const void *rows, const vnConv *g, const int64_t *row_list,
                                              int64_t row_cap, const int32_t *row_count, void *workspace,
                                              size_t workspace_bytes, int32_t *chunks, vnStream stream) {
    VN_CHECK_ARG(chunks && workspace && row_list && row_count && row_cap > 0);
    return wgrad_impl(src, rows, static_cast<float *>(workspace), g, 0, row_list, row_cap, workspace, workspace_bytes, stream,
                      chunks, row_count);
}

// One of the three bf16 products of an fp32x3 weight gradient whose operands are stored split (g->dtype VN_F32X3S, strides in
// 4-byte elements as for every VN_F32X3S tensor): pass 0 = hi(src) . hi(rows), 1 = lo(src) . hi(rows), 2 = hi(src) . lo(rows),
// each as a run of the bf16 nine-tap patch kernel over the halves where they lie (no [hi | lo] copies).  Only for the
// geometries of that kernel (3x3 taps, stride 1 in H / W, <= 64 channels, images >= 128 x 128: the 64-channel Conv3d layers,
// model.py:207-209); VN_EUNSUPPORTED otherwise.  Partials and *chunks as vn_conv_wgrad_partials (the caller lays the three
// passes' slabs one after the other and the unpack sums them like row chunks).
extern "C" int vn_conv_wgrad_partials_split_pass(const void *src, const void *rows, const vnConv *g, int32_t pass, void *workspace,
                                                 size_t workspace_bytes, int32_t *chunks, vnStream stream) {
    VN_CHECK_ARG(g && chunks && workspace && pass >= 0 && pass <= 2 && g->dtype == VN_F32X3S);
    if ((g->Cs & 7) || (g->Cr & 7)) return VN_EUNSUPPORTED;
    vnConv b = *g;
    b.dtype = VN_BF16;
    b.src_sB *= 2; b.src_sD *= 2; b.src_sH *= 2; b.src_sW *= 2;
    b.out_sB *= 2; b.out_sD *= 2; b.out_sH *= 2; b.out_sW *= 2;
    return wgrad_impl(src, rows, static_cast<float *>(workspace), &b, 0, nullptr, 0, workspace, workspace_bytes, stream, chunks,
                      nullptr, pass);
}

extern "C" int vn_conv_wgrad_rows(const void *src, const void *rows, float *dw_packed, const vnConv *g,
                                  const int64_t *row_list, int64_t n_rows, void *workspace, size_t workspace_bytes,
                                  vnStream stream) {
    if (!row_list || n_rows < 0) return VN_EINVAL;
    if (n_rows == 0) return VN_OK;
    return wgrad_impl(src, rows, dw_packed, g, 0, row_list, n_rows, workspace, workspace_bytes, stream);
}

static int wgrad_impl(const void *src, const void *rows, float *dw_packed, const vnConv *g, int32_t split,
                      const int64_t *row_list, int64_t n_rows, void *workspace, size_t workspace_bytes, vnStream stream,
                      int32_t *chunks_out, const int32_t *row_count, int split_pass) {
    const bool partial_only = chunks_out != nullptr;   // leave the chunk partials in the workspace, no reduction
    if (partial_only) dw_packed = static_cast<float *>(workspace);
    VN_CHECK_ARG(src && rows && dw_packed && g);
    VN_CHECK_ARG(g->B > 0 && g->Ds > 0 && g->Hs > 0 && g->Ws > 0 && g->Dr > 0 && g->Hr > 0 && g->Wr > 0);
    VN_CHECK_ARG(g->kD >= 1 && g->kH >= 1 && g->kW >= 1 && g->kD * g->kH * g->kW <= 65535);
    if (!row_list && (g->divD != 1 || g->divH != 1 || g->divW != 1)) return VN_EUNSUPPORTED;
    if (g->divD < 1 || g->divH < 1 || g->divW < 1) return VN_EINVAL;
    VN_CHECK_ARG(g->dtype == VN_BF16 || ((g->dtype == VN_F32 || g->dtype == VN_F32X3 || g->dtype == VN_F32X3S) && !split));
    const bool ps = g->dtype == VN_F32X3S;    // both operands stored split: 16-channel tiles, strides in whole 8-channel groups
    if (ps && ((g->Cs & 15) || (g->Cr & 15) || ((g->src_sB | g->src_sD | g->src_sH | g->src_sW | g->out_sB | g->out_sD | g->out_sH | g->out_sW) & 7)))
        return VN_EUNSUPPORTED;
    const bool f32 = g->dtype != VN_BF16;
    const int esz = f32 ? 4 : 2, al = 16 / esz - 1;
    if (g->Cs <= 0 || (g->Cs & al) || g->Cr <= 0 || (g->Cr & al)) return VN_EUNSUPPORTED;
    if (((g->src_sB | g->src_sD | g->src_sH | g->src_sW) & al) != 0) return VN_EUNSUPPORTED;
    if (((g->out_sB | g->out_sD | g->out_sH | g->out_sW) & al) != 0) return VN_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(rows) & 15)) return VN_EUNSUPPORTED;

    WGParams p{};
    p.src = static_cast<const char *>(src);
    p.rows = static_cast<const char *>(rows);
    p.dw = dw_packed;
    p.sB = g->src_sB; p.sD = g->src_sD; p.sH = g->src_sH; p.sW = g->src_sW;
    p.rB = g->out_sB; p.rD = g->out_sD; p.rH = g->out_sH; p.rW = g->out_sW;
    p.B = g->B; p.Ds = g->Ds; p.Hs = g->Hs; p.Ws = g->Ws; p.Dr = g->Dr; p.Hr = g->Hr; p.Wr = g->Wr;
    p.mulD = g->mulD; p.mulH = g->mulH; p.mulW = g->mulW;
    p.tmulD = g->tmulD; p.tmulH = g->tmulH; p.tmulW = g->tmulW;
    p.padD = g->padD; p.padH = g->padH; p.padW = g->padW;
    p.kD = g->kD; p.kH = g->kH; p.kW = g->kW;
    p.C = g->Cs; p.N = g->Cr;
    if (split) return VN_EUNSUPPORTED;      // (the [hi|lo] three-pass form of the retired bf16x3 mode: round 5)
    p.x3 = g->dtype == VN_F32X3 || ps;
    // split_pass >= 0: g describes ONE bf16 half of two split-storage tensors (strides in bf16 elements = twice the 4-byte
    // ones, a site's channels spread over 2 C bf16 slots): only the patch form reads that layout
    const int wmul = split_pass >= 0 ? 2 : 1;
    const int64_t sbytes = ((int64_t)(g->B - 1) * g->src_sB + (int64_t)(g->Ds - 1) * g->src_sD +
                            (int64_t)(g->Hs - 1) * g->src_sH + (int64_t)(g->Ws - 1) * g->src_sW + wmul * g->Cs) * esz;
    const int64_t rbytes = row_list ? ((n_rows - 1) * g->out_sW + wmul * g->Cr) * esz
                                    : ((int64_t)(g->B - 1) * g->out_sB + (int64_t)(g->Dr - 1) * g->out_sD +
                                       (int64_t)(g->Hr - 1) * g->out_sH + (int64_t)(g->Wr - 1) * g->out_sW + wmul * g->Cr) * esz;
    if (sbytes > (int64_t)WG_MAX_WINDOW || rbytes > (int64_t)WG_MAX_WINDOW) return VN_EUNSUPPORTED;
    p.src_bytes = (uint32_t)sbytes;
    p.rows_bytes = (uint32_t)rbytes;

    const int64_t M = row_list ? n_rows : (int64_t)g->B * g->Dr * g->Hr * g->Wr;
    const WGPlan w = wgrad_plan(g, split, M);
    const bool tri = w.tri, n128 = w.n128, k128 = w.k128;
    const int tiles_n = w.tiles_n, tiles_k = w.tiles_k, groups = w.groups;
    p.tiles_k = tiles_k;
    p.row_list = row_list;
    p.n_rows = n_rows;
    p.row_count = row_list ? row_count : nullptr;
    p.divD = g->divD; p.divH = g->divH; p.divW = g->divW;
    const WPPlan wp = wgrad_patch_plan(g, split, row_list != nullptr);
    if (wp.ok && workspace && !(reinterpret_cast<uintptr_t>(workspace) & 15) && !(w.dw_elems & 3) &&
        (int64_t)(workspace_bytes / ((size_t)w.dw_elems * sizeof(float))) >= 1) {
        int64_t pchunks = wp.chunks;
        const int64_t room = (int64_t)(workspace_bytes / ((size_t)w.dw_elems * sizeof(float)));
        if (pchunks > room) pchunks = room;
        WPParams q{};
        q.src = p.src; q.rows = p.rows;
        q.part = static_cast<float *>(workspace);
        q.part_stride = w.dw_elems;
        q.sB = p.sB; q.sD = p.sD; q.sH = p.sH; q.sW = p.sW;
        q.rB = p.rB; q.rD = p.rD; q.rH = p.rH; q.rW = p.rW;
        q.B = p.B; q.Ds = p.Ds; q.Hs = p.Hs; q.Ws = p.Ws; q.Dr = p.Dr; q.Hr = p.Hr; q.Wr = p.Wr;
        q.mulD = p.mulD; q.padD = p.padD; q.kD = p.kD;
        q.C = p.C; q.N = p.N; q.tiles_k = wp.tiles_k;
        q.tiles_x = wp.tiles_x; q.tiles_y = wp.tiles_y;
        q.tiles_per_chunk = (int32_t)vn_ceil_div(wp.ntiles, pchunks);
        pchunks = vn_ceil_div(wp.ntiles, q.tiles_per_chunk);
        q.src_bytes = p.src_bytes; q.rows_bytes = p.rows_bytes;
        q.s_cb = q.r_cb = split_pass >= 0 ? 32u : 16u;
        q.s_part = split_pass == 1 ? 16u : 0u;      // pass 1: lo(src) . hi(rows)
        q.r_part = split_pass == 2 ? 16u : 0u;      // pass 2: hi(src) . lo(rows)
        hipStream_t pst = vn_stream(stream);
        const dim3 pgrid((unsigned)pchunks, (unsigned)g->kD, (unsigned)(wp.tiles_n * wp.tiles_k));
        {
            constexpr size_t lds = 2 * (64 * 128 + 6 * 32 * 128);
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_patch<1>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (attr != hipSuccess) return (int)attr;
            k_wgrad_patch<1><<<pgrid, 256, lds, pst>>>(q);
        }
        VN_LAUNCH_STATUS();
        if (partial_only) {
            *chunks_out = (int32_t)pchunks;
        } else {
            const int64_t n4 = w.dw_elems / 4;
            int64_t blocks = vn_ceil_div(n4, 256);
            if (blocks > 2048) blocks = 2048;
            k_wgrad_reduce<<<(unsigned)blocks, 256, 0, pst>>>(q.part, (int)pchunks, n4, dw_packed);
            VN_LAUNCH_STATUS();
        }
        return VN_OK;
    }
    if (split_pass >= 0) return VN_EUNSUPPORTED;      // (geometries outside the patch form)
    // as many row chunks as the workspace has room for partial sums (none: one chunk, accumulated in place)
    int64_t chunks = w.chunks;
    const int64_t room = workspace ? (int64_t)(workspace_bytes / ((size_t)w.dw_elems * sizeof(float))) : 0;
    if (chunks > room) chunks = room;
    if ((w.dw_elems & 3) || (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(dw_packed) & 15))
        chunks = 1;
    if (chunks < 2) chunks = 1;
    int64_t rpc = vn_ceil_div(vn_ceil_div(M, chunks), 64) * 64;
    if (rpc < 64) rpc = 64;
    if (rpc > (1 << 30)) return VN_EUNSUPPORTED;
    p.rows_per_chunk = (int32_t)rpc;
    chunks = vn_ceil_div(M, rpc);
    p.part = (chunks > 1 || partial_only) ? static_cast<float *>(workspace) : nullptr;
    if (partial_only) {
        if (!workspace || (int64_t)(workspace_bytes / ((size_t)w.dw_elems * sizeof(float))) < chunks) return VN_EWORKSPACE;
        *chunks_out = (int32_t)chunks;
    }
    p.part_stride = w.dw_elems;
    dim3 grid((unsigned)chunks, (unsigned)groups, (unsigned)(tiles_n * tiles_k));
    hipStream_t st = vn_stream(stream);
    int rc;
    if (tri) rc = k128 ? launch_wgrad<2, 2, false, 3, 4>(p, grid, st) : launch_wgrad<2, 1, false, 3, 4>(p, grid, st);
    else if (ps) {
        if (n128 && k128) rc = launch_wgrad<4, 2, true, 1, 4, true>(p, grid, st);
        else if (n128) rc = launch_wgrad<4, 1, true, 1, 4, true>(p, grid, st);
        else if (k128) rc = launch_wgrad<2, 2, true, 1, 4, true>(p, grid, st);
        else rc = launch_wgrad<2, 1, true, 1, 4, true>(p, grid, st);
    } else if (f32 && p.x3) {
        // fp32x3: a stage is 48 bf16 MFMAs per 128 x 128 tile instead of 128 fp32 ones — the four-wave workgroup that the
        // exact fp32 path can afford (its MFMAs hide everything) leaves the fragment gathers exposed: eight waves
        if (n128 && k128) rc = launch_wgrad<4, 2, true, 1, 4>(p, grid, st);
        else if (n128) rc = launch_wgrad<4, 1, true, 1, 4>(p, grid, st);
        else if (k128) rc = launch_wgrad<2, 2, true, 1, 4>(p, grid, st);
        else rc = launch_wgrad<2, 1, true, 1, 4>(p, grid, st);
    } else if (f32) {
        if (n128 && k128) rc = launch_wgrad<4, 4, true, 1>(p, grid, st);
        else if (n128) rc = launch_wgrad<4, 2, true, 1>(p, grid, st);
        else if (k128) rc = launch_wgrad<2, 4, true, 1>(p, grid, st);
        else rc = launch_wgrad<2, 2, true, 1>(p, grid, st);
    } else {
        // eight waves on the 128 x 128 tile (two per SIMD at one workgroup per CU; the four-wave form of round 2 is gone)
        if (n128 && k128) rc = launch_wgrad<4, 2, false, 1, 4>(p, grid, st);
        else if (n128) rc = launch_wgrad<4, 1, false, 1, 4>(p, grid, st);
        else if (k128) rc = launch_wgrad<2, 2, false, 1, 4>(p, grid, st);
        else rc = launch_wgrad<2, 1, false, 1, 4>(p, grid, st);
    }
    if (rc != VN_OK) return rc;
    if (chunks > 1 && !partial_only) {
        const int64_t n4 = w.dw_elems / 4;
        int64_t blocks = vn_ceil_div(n4, 256);
        if (blocks > 2048) blocks = 2048;
        k_wgrad_reduce<<<(unsigned)blocks, 256, 0, st>>>(p.part, (int)chunks, n4, dw_packed);
        VN_LAUNCH_STATUS();
    }
    return VN_OK;
}
