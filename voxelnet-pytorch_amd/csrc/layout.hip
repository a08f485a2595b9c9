// Weight packing and layout/dtype helpers at the nn.Module boundary.
//   vn_pack_weight / vn_unpack_wgrad : torch parameter layouts (model.py:134-153,188-193)
//                                      <-> the gather-GEMM's [tap][N][K] bf16 operand
//   vn_nchw_to_rows / vn_rows_to_nchw: NC(D)HW fp32 <-> channels-last rows (model.py:259,281)
//   vn_cast_rows, vn_col_sums, vn_heads_bwd
// All HBM-bound elementwise/transposing kernels: 16-B accesses on the channels-last side,
// LDS-tiled transposes so both sides stay coalesced.
#include "common.h"

namespace {

// torch index of packed (tap, n, k) ; mode: 0/1 conv (Cout,Cin,taps), 2/3 convT (Cin,Cout,taps)
__device__ __forceinline__ int64_t torch_index(int mode, int c_out, int c_in, int taps, int tap, int n, int k,
                                               int cin_fold) {
    // modes 0,2: n = cout, k = cin ; modes 1,3: n = cin, k = cout
    int co = (mode == 0 || mode == 2) ? n : k;
    int ci = (mode == 0 || mode == 2) ? k : n;
    if (cin_fold > 1) {
        const int per = c_in / cin_fold;
        ci = (ci % per) * cin_fold + ci / per;
    }
    if (mode <= 1) return ((int64_t)co * c_in + ci) * taps + tap;
    return ((int64_t)ci * c_out + co) * taps + tap;
}

// packed_dtype VN_F32X3 (include/voxelnet_hip.h, vn_pack_weight): element (row, k) of a [rows][K] fp32-sized operand, K % 32
// == 0 -> the "split fp32" storage format (vnDtype VN_F32X3S, round 5): every group of 8 channels is 32 B = its eight hi
// bf16 parts, then its eight lo parts.  A lane (fq = lane >> 4) of the fp32x3 convolution kernels reads the two 16-B
// granules 2 fq and 2 fq + 1 of every 128-B chunk: the hi and lo parts of channels 8 fq .. 8 fq + 7, its eight k values of
// one v_mfma_f32_16x16x32_bf16 (the same channels vn_split8 takes from the two fp32 granules of an unsplit operand)
__device__ __forceinline__ void store_x3_weight(void *packed, int64_t i, int K, float v) {
    const int k = (int)(i % K);
    const int pos = k & 7;
    bf16_t *group = static_cast<bf16_t *>(packed) + (i - pos) * 2;      // 16 bf16 slots per 8-channel group
    bf16_t hi, lo;
    vn_split_bf16(v, hi, lo);
    group[pos] = hi;
    group[8 + pos] = lo;
}

__global__ void __launch_bounds__(256) k_pack_weight(const float *__restrict__ w, int c_out, int c_in, int taps,
                                                     int mode, int split3, int cin_fold, void *__restrict__ packed, int f32) {
    const int N = (mode == 0 || mode == 2) ? c_out : c_in;
    const int K = (mode == 0 || mode == 2) ? c_in : c_out;
    const int Ke = split3 ? 3 * K : K;
    const int64_t total = (int64_t)taps * N * Ke;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ke = (int)(i % Ke);
        const int n = (int)((i / Ke) % N);
        const int tap = (int)(i / ((int64_t)Ke * N));
        const int k = ke % K, part = ke / K;   // part 0: hi, 1: hi, 2: lo
        const float v = w[torch_index(mode, c_out, c_in, taps, tap, n, k, cin_fold)];
        if (f32 == 2) {
            store_x3_weight(packed, i, K, v);
        } else if (f32) {
            static_cast<float *>(packed)[i] = v;
        } else {
            bf16_t hi, lo;
            vn_split_bf16(v, hi, lo);
            static_cast<bf16_t *>(packed)[i] = part == 2 ? lo : hi;
        }
    }
}

__global__ void __launch_bounds__(256) k_unpack_wgrad(const float *__restrict__ dwp, int c_out, int c_in, int taps,
                                                      int mode, int cin_fold, float *__restrict__ dw) {
    // dwp is [tap][N = cout][K = cin] (forward orientation) for both conv (mode 0) and convT (mode 2)
    const int64_t total = (int64_t)taps * c_out * c_in;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % c_in);
        const int n = (int)((i / c_in) % c_out);
        const int tap = (int)(i / ((int64_t)c_in * c_out));
        dw[torch_index(mode, c_out, c_in, taps, tap, n, k, cin_fold)] = dwp[i];
    }
}

constexpr int UNPACK_LDS_FLOATS = 8192;   // 32 KB: the LDS tile of the batched pack / unpack kernels
// ---- batched forms: every layer of the network in ONE launch (48 + 24 launches of ~5 us of dispatch each otherwise)
constexpr int PACK_MAX_JOBS = 56;
struct PackJobs {
    int32_t n;
    int32_t x3_presplit;                // VN_F32X3 jobs with K % 32 == 0: hi / lo granules (vn_x3_presplit)
    int32_t block_begin[PACK_MAX_JOBS + 1];
    int32_t tiled[PACK_MAX_JOBS];       // 0: one thread per element; 1: row tiles (modes 0, 3); 2: 32 x 8 tiles (modes 1, 2)
    vnPackJob job[PACK_MAX_JOBS];
};
struct UnpackJobs {
    int32_t n;
    int32_t block_begin[PACK_MAX_JOBS + 1];
    int32_t tn[PACK_MAX_JOBS];          // mode 0: torch rows (n) per workgroup tile
    vnUnpackJob job[PACK_MAX_JOBS];
};

__device__ __forceinline__ int find_job(const int32_t *block_begin, int n) {
    int j = 0;
    while (j + 1 < n && (int)blockIdx.x >= block_begin[j + 1]) ++j;   // block-uniform (scalar) search
    return j;
}

// The mirror image of k_unpack_wgrads_batch: one thread per packed element reads the torch tensor 4 bytes at a time,
// `taps` floats apart.  bf16 jobs without the three-part split go through LDS instead — a workgroup reads whole contiguous
// runs of the torch layout and writes contiguous packed rows:
//   row tiles  (modes 0 and 3: the packed row index n is the torch tensor's OUTER index): two n per workgroup — their
//              (inner, tap) runs -> LDS -> `taps` packed rows of K bf16 each;
//   32 x 8 tiles (modes 1 and 2, no fold: n is the torch tensor's MIDDLE index): 32 k x 8 n x taps — per k a run of
//              8 x taps floats -> LDS -> per (tap, n) 32 consecutive bf16.
__global__ void __launch_bounds__(256) k_pack_weights_batch(const PackJobs t) {
    __shared__ __attribute__((aligned(16))) float tile[UNPACK_LDS_FLOATS];
    {
        const int j0 = find_job(t.block_begin, t.n);
        const vnPackJob &q = t.job[j0];
        const int b = blockIdx.x - t.block_begin[j0];
        const int taps = q.taps;
        bf16_t *out = static_cast<bf16_t *>(q.packed);
        if (t.tiled[j0] == 1) {
            // mode 0: n = co, inner = ci (folded), K = c_in; mode 3: n = ci (folded), inner = co, K = c_out
            const int N = q.mode == 0 ? q.c_out : q.c_in, K = q.mode == 0 ? q.c_in : q.c_out;
            const int per = q.c_in / q.cin_fold;
            const int row_floats = K * taps;
            const int rows = 2 * b + 2 <= N ? 2 : N - 2 * b;
            for (int r = 0; r < rows; ++r) {
                const int n = 2 * b + r;
                // torch outer index of packed row n: mode 0 -> co = n; mode 3 -> ci = fold(n)
                const int outer = q.mode == 0 ? n : (q.cin_fold > 1 ? (n % per) * q.cin_fold + n / per : n);
                const float *src = q.w + (int64_t)outer * row_floats;
                for (int e = threadIdx.x; e < row_floats; e += 256) tile[r * row_floats + e] = src[e];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < rows * taps * K; e += 256) {
                const int k = e % K, tap = (e / K) % taps, r = e / (K * taps);
                // torch inner index of packed column k: mode 0 -> ci = fold(k); mode 3 -> co = k
                const int inner = (q.mode == 0 && q.cin_fold > 1) ? (k % per) * q.cin_fold + k / per : k;
                out[((int64_t)tap * N + 2 * b + r) * K + k] = (bf16_t)tile[r * row_floats + inner * taps + tap];
            }
            return;
        }
        if (t.tiled[j0] == 2) {
            // mode 1: n = ci, k = co, torch[(k c_in + n) taps + tap]; mode 2: n = co, k = ci, torch[(k c_out + n) taps + tap]
            const int N = q.mode == 2 ? q.c_out : q.c_in, K = q.mode == 2 ? q.c_in : q.c_out;
            const int tiles_n = (N + 7) >> 3, tkb = b / tiles_n, tnb = b - tkb * tiles_n;
            const int k0 = tkb * 32, n0 = tnb * 8;
            const int kk = k0 + 32 <= K ? 32 : K - k0, nn = n0 + 8 <= N ? 8 : N - n0;
            const int run = nn * taps;
            for (int e = threadIdx.x; e < kk * run; e += 256) {
                const int kl = e / run, r = e - kl * run;
                tile[e] = q.w[((int64_t)(k0 + kl) * N + n0) * taps + r];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < taps * nn * kk; e += 256) {
                const int kl = e % kk, nl = (e / kk) % nn, tap = e / (kk * nn);
                out[((int64_t)tap * N + n0 + nl) * K + k0 + kl] = (bf16_t)tile[kl * run + nl * taps + tap];
            }
            return;
        }
    }
    const int j = find_job(t.block_begin, t.n);
    const vnPackJob &q = t.job[j];
    const int nb = t.block_begin[j + 1] - t.block_begin[j], b = blockIdx.x - t.block_begin[j];
    const int N = (q.mode == 0 || q.mode == 2) ? q.c_out : q.c_in;
    const int K = (q.mode == 0 || q.mode == 2) ? q.c_in : q.c_out;
    const int Ke = q.split3 ? 3 * K : K;
    const int64_t total = (int64_t)q.taps * N * Ke;
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < total; i += (int64_t)nb * 256) {
        const int ke = (int)(i % Ke);
        const int n = (int)((i / Ke) % N);
        const int tap = (int)(i / ((int64_t)Ke * N));
        const int k = ke % K, part = ke / K;
        const float v = q.w[torch_index(q.mode, q.c_out, q.c_in, q.taps, tap, n, k, q.cin_fold)];
        if (q.packed_dtype == VN_F32X3 && t.x3_presplit && (K & 31) == 0) {
            store_x3_weight(q.packed, i, K, v);
        } else if (q.packed_dtype != VN_BF16) {
            static_cast<float *>(q.packed)[i] = v;
        } else {
            bf16_t hi, lo;
            vn_split_bf16(v, hi, lo);
            static_cast<bf16_t *>(q.packed)[i] = part == 2 ? lo : hi;
        }
    }
}

// The torch layout keeps the taps of one (co, ci) pair together, the packed layout keeps one tap's (n, k) matrix together:
// a thread-per-element unpack stores 4-byte words `taps` floats apart (partial-line writes; the launch cost 2 % of the
// train step).  So a workgroup owns a TILE of the torch layout and goes through LDS: coalesced (summed) reads of the
// packed rows, then one contiguous run of stores per torch row.
//   mode 0 (torch[(n c_in + fold(k)) taps + tap]): tn whole n rows (every k, every tap: the fold permutes k inside a row);
//   mode 2 (torch[(k c_out + n) taps + tap], fold 1): 16 n x 16 k, every tap — 16 runs of 16 x taps floats.
// block_begin counts tiles; t.tn[j] = rows per tile of job j (mode 0).

__device__ __forceinline__ float4 unpack_sum(const vnUnpackJob &q, int64_t i) {
    const int chunks = q.chunks > 1 ? q.chunks : 1;
    float4 v = *reinterpret_cast<const float4 *>(q.dw_packed + i);
    int c = 1;
    for (; c + 8 <= chunks; c += 8) {                                 // eight partial slabs in flight, added in order
        float4 u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = *reinterpret_cast<const float4 *>(q.dw_packed + (int64_t)(c + e) * q.chunk_stride + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v.x += u[e].x; v.y += u[e].y; v.z += u[e].z; v.w += u[e].w; }
    }
    for (; c < chunks; ++c) {                                         // fixed order
        const float4 u = *reinterpret_cast<const float4 *>(q.dw_packed + (int64_t)c * q.chunk_stride + i);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    return v;
}

__global__ void __launch_bounds__(256) k_unpack_wgrads_batch(const UnpackJobs t) {
    __shared__ __attribute__((aligned(16))) float tile[UNPACK_LDS_FLOATS];
    const int j = find_job(t.block_begin, t.n);
    const vnUnpackJob &q = t.job[j];
    const int b = blockIdx.x - t.block_begin[j];
    const int taps = q.taps, cin = q.c_in, cout = q.c_out;
    if (q.mode == 0) {
        const int tn = t.tn[j], n0 = b * tn, rows = n0 + tn <= cout ? tn : cout - n0;
        const int k4n = cin >> 2, row_floats = cin * taps, per = cin / q.cin_fold;
        // packed -> LDS in torch order: tile[nl][fold(k) * taps + tap]
        for (int e = threadIdx.x; e < taps * rows * k4n; e += 256) {
            const int k4 = e % k4n, nl = (e / k4n) % rows, tap = e / (k4n * rows);
            const float4 v = unpack_sum(q, ((int64_t)tap * cout + n0 + nl) * cin + k4 * 4);
            const float ev[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int k = k4 * 4 + d, ci = q.cin_fold > 1 ? (k % per) * q.cin_fold + k / per : k;
                tile[nl * row_floats + ci * taps + tap] = ev[d];
            }
        }
        __syncthreads();
        float *dst = q.dw + (int64_t)n0 * row_floats;      // rows n0 .. n0 + rows - 1 are one contiguous run
        for (int e = threadIdx.x; e < rows * row_floats; e += 256) dst[e] = tile[e];
    } else {
        const int tiles_k = (cin + 15) >> 4, tnb = b / tiles_k, tkb = b - tnb * tiles_k;
        const int n0 = tnb * 16, k0 = tkb * 16;
        const int nn = n0 + 16 <= cout ? 16 : cout - n0, kk = k0 + 16 <= cin ? 16 : cin - k0;   // (cin % 4 == 0)
        const int run = nn * taps;                                   // floats of one k's run inside this tile
        // packed -> LDS: tile[kl][nl * taps + tap]
        for (int e = threadIdx.x; e < taps * nn * (kk >> 2); e += 256) {
            const int k4 = e % (kk >> 2), nl = (e / (kk >> 2)) % nn, tap = e / ((kk >> 2) * nn);
            const float4 v = unpack_sum(q, ((int64_t)tap * cout + n0 + nl) * cin + k0 + k4 * 4);
            const float ev[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) tile[(k4 * 4 + d) * run + nl * taps + tap] = ev[d];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kk * run; e += 256) {
            const int kl = e / run, r = e - kl * run;
            q.dw[((int64_t)(k0 + kl) * cout + n0) * taps + r] = tile[e];
        }
    }
}

__device__ __forceinline__ void store_elem(void *dst, int dtype, int64_t lo_off, int64_t row_off, int c, float v) {
    if (dtype == VN_F32) {
        static_cast<float *>(dst)[row_off + c] = v;
    } else {
        bf16_t hi, lo;
        vn_split_bf16(v, hi, lo);
        static_cast<bf16_t *>(dst)[row_off + c] = hi;
        if (lo_off) static_cast<bf16_t *>(dst)[row_off + lo_off + c] = lo;
    }
}

// (B,C,S) fp32 -> rows (B*S, C): 32x32 LDS-tiled transpose
__global__ void __launch_bounds__(256) k_nchw_to_rows(const float *__restrict__ src, int C, int64_t S, void *dst,
                                                      int dtype, int64_t dst_stride, int64_t lo_off) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const int64_t s = s0 + tx;
        tile[j][tx] = (c < C && s < S) ? src[((int64_t)b * C + c) * S + s] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t s = s0 + j;
        const int c = c0 + tx;
        if (c < C && s < S) store_elem(dst, dtype, lo_off, ((int64_t)b * S + s) * dst_stride, c, tile[tx][j]);
    }
}

__global__ void __launch_bounds__(256) k_rows_to_nchw(const void *__restrict__ src, int dtype, int64_t src_stride,
                                                      int C, int64_t S, float *__restrict__ dst, int sigmoid_first_n) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int64_t s = s0 + j;
        const int c = c0 + tx;
        float v = 0.f;
        if (c < C && s < S) {
            const int64_t o = ((int64_t)b * S + s) * src_stride + c;
            v = dtype == VN_F32 ? static_cast<const float *>(src)[o] : (float)static_cast<const bf16_t *>(src)[o];
            if (c < sigmoid_first_n) v = 1.0f / (1.0f + expf(-v));
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const int64_t s = s0 + tx;
        if (c < C && s < S) dst[((int64_t)b * C + c) * S + s] = tile[tx][j];
    }
}

// heads epilogue in one launch (model.py:276-281): rows (B*S,16) fp32 = [2 prob logits | 14 reg] -> prob = sigmoid
// (B,2,S) and reg (B,14,S), NCHW.  One thread per site: a 64-B row read, 16 stores that are coalesced across the wave
// (consecutive sites of one channel).
__global__ void __launch_bounds__(256) k_heads_to_nchw(const float *__restrict__ rows, int B, int64_t S, float *__restrict__ prob,
                                                       float *__restrict__ reg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * S) return;
    const int64_t b = i / S, s = i - b * S;
    const float4 *r = reinterpret_cast<const float4 *>(rows + i * 16);
    const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
    const float v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    prob[(b * 2 + 0) * S + s] = 1.0f / (1.0f + expf(-v[0]));
    prob[(b * 2 + 1) * S + s] = 1.0f / (1.0f + expf(-v[1]));
#pragma unroll
    for (int c = 0; c < 14; ++c) reg[(b * 14 + c) * S + s] = v[2 + c];
}

// 4 channels per thread
__global__ void __launch_bounds__(256) k_cast_rows(const void *__restrict__ src, int sdt, int64_t sstride, int64_t M,
                                                   int C, void *__restrict__ dst, int ddt, int64_t dstride, int64_t lo_off) {
    const int groups = C >> 2;
    const int64_t total = M * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / groups;
        const int c = (int)(i - m * groups) << 2;
        float v[4];
        if (sdt == VN_F32X3S) {      // split fp32 storage (per 8 channels: eight hi bf16 parts, then eight lo parts)
            const bf16_t *g = reinterpret_cast<const bf16_t *>(static_cast<const float *>(src) + m * sstride + (c & ~7)) + (c & 7);
            const bf16x4_t hi = *reinterpret_cast<const bf16x4_t *>(g), lo = *reinterpret_cast<const bf16x4_t *>(g + 8);
            if (ddt == VN_BF16 && lo_off) {      // -> [hi | lo] bf16 rows: the parts as they are
                bf16_t *d = static_cast<bf16_t *>(dst) + m * dstride + c;
                *reinterpret_cast<bf16x4_t *>(d) = hi;
                *reinterpret_cast<bf16x4_t *>(d + lo_off) = lo;
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (float)hi[j] + (float)lo[j];
        } else if (sdt == VN_F32) {
            const float4 t = *reinterpret_cast<const float4 *>(static_cast<const float *>(src) + m * sstride + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const bf16x4_t t = *reinterpret_cast<const bf16x4_t *>(static_cast<const bf16_t *>(src) + m * sstride + c);
            v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
        }
        if (ddt == VN_F32) {
            *reinterpret_cast<float4 *>(static_cast<float *>(dst) + m * dstride + c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            bf16x4_t hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bf16_t h, l; vn_split_bf16(v[j], h, l); hi[j] = h; lo[j] = l; }
            bf16_t *d = static_cast<bf16_t *>(dst) + m * dstride + c;
            *reinterpret_cast<bf16x4_t *>(d) = hi;
            if (lo_off) *reinterpret_cast<bf16x4_t *>(d + lo_off) = lo;
        }
    }
}

// column sums, pass 1: thread (cg = tid % groups) walks rows; LDS reduce; one partial row per workgroup (plain stores:
// no atomics, so the result does not depend on the order in which workgroups finish)
__global__ void __launch_bounds__(256) k_col_sums(const void *__restrict__ rows, int dtype, int64_t stride, int64_t M,
                                                  int C, float *__restrict__ part /* [gridDim.x][C] */) {
    __shared__ float red[256 * 4];
    const int groups = C >> 2;            // <= 256
    const int rpb = 256 / groups;         // rows per block iteration
    const int cg = threadIdx.x % groups, rr = threadIdx.x / groups;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (rr < rpb) {
        for (int64_t m = (int64_t)blockIdx.x * rpb + rr; m < M; m += (int64_t)gridDim.x * rpb) {
            if (dtype == VN_F32) {
                const float4 t = *reinterpret_cast<const float4 *>(static_cast<const float *>(rows) + m * stride + cg * 4);
                s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
            } else {
                const bf16x4_t t = *reinterpret_cast<const bf16x4_t *>(static_cast<const bf16_t *>(rows) + m * stride + cg * 4);
                s[0] += (float)t[0]; s[1] += (float)t[1]; s[2] += (float)t[2]; s[3] += (float)t[3];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x * 4 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < groups) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rpb; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] += red[(r * groups + threadIdx.x) * 4 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(size_t)blockIdx.x * C + threadIdx.x * 4 + j] = t[j];
    }
}

// pass 2: out[c] = sum of the partial rows in a fixed order (double)
__global__ void __launch_bounds__(256) k_col_sums_final(const float *__restrict__ part, int nblk, int C, float *__restrict__ out) {
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)part[(size_t)b * C + c];
        out[c] = (float)s;
    }
}

// heads backward: NCHW d_prob (B,2,S), d_reg (B,14,S), prob -> rows (B*S, 16)
__global__ void __launch_bounds__(256) k_heads_bwd(const float *__restrict__ dprob, const float *__restrict__ dreg,
                                                   const float *__restrict__ prob, int B, int64_t S, void *drows,
                                                   int64_t stride, int split, int f32) {
    const int64_t total = (int64_t)B * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / S, s = i - b * S;
        bf16_t *d = static_cast<bf16_t *>(drows) + i * stride;
        float *df = static_cast<float *>(drows) + i * stride;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v;
            if (c < 2) {
                const float pr = prob[(b * 2 + c) * S + s];
                v = dprob[(b * 2 + c) * S + s] * pr * (1.0f - pr);
            } else {
                v = dreg[(b * 14 + (c - 2)) * S + s];
            }
            if (f32) {
                df[c] = v;
            } else {
                bf16_t h, l;
                vn_split_bf16(v, h, l);
                d[c] = h;
                if (split) d[16 + c] = l;
            }
        }
    }
}

inline unsigned gs_blocks(int64_t total, int per_block = 256, int cap = 8192) {
    int64_t b = vn_ceil_div(total, per_block);
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

extern "C" int vn_pack_weight(const float *w, int32_t c_out, int32_t c_in, int32_t taps, int32_t mode, int32_t split3,
                              int32_t cin_fold, void *packed, vnDtype packed_dtype, vnStream stream) {
    VN_CHECK_ARG(w && packed && c_out > 0 && c_in > 0 && taps > 0 && mode >= 0 && mode <= 3);
    VN_CHECK_ARG(packed_dtype == VN_BF16 || packed_dtype == VN_F32 || packed_dtype == VN_F32X3);
    if (split3) return VN_EUNSUPPORTED;      // ([hi;hi;lo] K expansion of the retired bf16x3 mode: round 5)
    VN_CHECK_ARG(cin_fold >= 1 && c_in % cin_fold == 0);
    const int64_t total = (int64_t)taps * c_out * c_in * (split3 ? 3 : 1);
    const int K = (mode == 0 || mode == 2) ? c_in : c_out;
    k_pack_weight<<<gs_blocks(total), 256, 0, vn_stream(stream)>>>(w, c_out, c_in, taps, mode, split3, cin_fold, packed,
                                                                   packed_dtype == VN_BF16 ? 0 : (packed_dtype == VN_F32X3 && vn_x3_presplit(K)) ? 2 : 1);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_unpack_wgrad(const float *dw_packed, int32_t c_out, int32_t c_in, int32_t taps, int32_t mode,
                               int32_t cin_fold, float *dw, vnStream stream) {
    VN_CHECK_ARG(dw_packed && dw && c_out > 0 && c_in > 0 && taps > 0 && (mode == 0 || mode == 2));
    VN_CHECK_ARG(cin_fold >= 1 && c_in % cin_fold == 0);
    const int64_t total = (int64_t)taps * c_out * c_in;
    k_unpack_wgrad<<<gs_blocks(total), 256, 0, vn_stream(stream)>>>(dw_packed, c_out, c_in, taps, mode, cin_fold, dw);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_pack_weights_batch(const vnPackJob *jobs, int32_t n, vnStream stream) {
    VN_CHECK_ARG(n >= 0 && (jobs || n == 0));
    for (int32_t base = 0; base < n; base += PACK_MAX_JOBS) {
        PackJobs t{};
        t.n = n - base < PACK_MAX_JOBS ? n - base : PACK_MAX_JOBS;
        t.x3_presplit = vn_x3_presplit(32);
        int blocks = 0;
        for (int j = 0; j < t.n; ++j) {
            const vnPackJob &q = jobs[base + j];
            VN_CHECK_ARG(q.w && q.packed && q.c_out > 0 && q.c_in > 0 && q.taps > 0 && q.mode >= 0 && q.mode <= 3);
            VN_CHECK_ARG(q.packed_dtype == VN_BF16 || q.packed_dtype == VN_F32 || q.packed_dtype == VN_F32X3);
            if (q.split3) return VN_EUNSUPPORTED;      // (retired with the bf16x3 mode: round 5)
            VN_CHECK_ARG(q.cin_fold >= 1 && q.c_in % q.cin_fold == 0);
            const int64_t total = (int64_t)q.taps * q.c_out * q.c_in * (q.split3 ? 3 : 1);
            int64_t nb = vn_ceil_div(total, 256 * 8);          // 8 elements per thread
            if (nb > 256) nb = 256;
            t.tiled[j] = 0;
            if (q.packed_dtype == VN_BF16 && !q.split3) {
                const int N = (q.mode == 0 || q.mode == 2) ? q.c_out : q.c_in;
                const int K = (q.mode == 0 || q.mode == 2) ? q.c_in : q.c_out;
                if ((q.mode == 0 || q.mode == 3) && (int64_t)2 * K * q.taps <= 8192) {
                    t.tiled[j] = 1;
                    nb = vn_ceil_div(N, 2);
                } else if ((q.mode == 1 || q.mode == 2) && q.cin_fold == 1 && (int64_t)32 * 8 * q.taps <= 8192) {
                    t.tiled[j] = 2;
                    nb = vn_ceil_div(K, 32) * vn_ceil_div(N, 8);
                }
            }
            t.job[j] = q;
            t.block_begin[j] = blocks;
            blocks += (int)nb;
        }
        t.block_begin[t.n] = blocks;
        k_pack_weights_batch<<<blocks, 256, 0, vn_stream(stream)>>>(t);
        VN_LAUNCH_STATUS();
    }
    return VN_OK;
}

extern "C" int vn_unpack_wgrads_batch(const vnUnpackJob *jobs, int32_t n, vnStream stream) {
    VN_CHECK_ARG(n >= 0 && (jobs || n == 0));
    for (int32_t base = 0; base < n; base += PACK_MAX_JOBS) {
        UnpackJobs t{};
        t.n = n - base < PACK_MAX_JOBS ? n - base : PACK_MAX_JOBS;
        int blocks = 0;
        for (int j = 0; j < t.n; ++j) {
            const vnUnpackJob &q = jobs[base + j];
            VN_CHECK_ARG(q.dw_packed && q.dw && q.c_out > 0 && q.c_in > 0 && q.taps > 0 && (q.mode == 0 || q.mode == 2));
            VN_CHECK_ARG(q.cin_fold >= 1 && q.c_in % q.cin_fold == 0);
            VN_CHECK_ARG(q.chunks <= 1 || q.chunk_stride >= (int64_t)q.taps * q.c_out * q.c_in);
            VN_CHECK_ARG((q.c_in & 3) == 0 && (q.chunk_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(q.dw_packed) & 15) == 0);
            int64_t nb;
            if (q.mode == 0) {   // whole torch rows per tile, as many as fit the LDS tile (at most 2)
                const int64_t row_floats = (int64_t)q.c_in * q.taps;
                if (row_floats > 8192) return VN_EUNSUPPORTED;
                int tn = (int)(8192 / row_floats);
                if (tn > 2) tn = 2;          // (latency-bound: many small tiles — 2 rows measured against 8)
                t.tn[j] = tn;
                nb = vn_ceil_div(q.c_out, tn);
            } else {
                if (q.cin_fold != 1 || (int64_t)16 * 16 * q.taps > 8192) return VN_EUNSUPPORTED;
                t.tn[j] = 16;
                nb = vn_ceil_div(q.c_out, 16) * vn_ceil_div(q.c_in, 16);
            }
            t.job[j] = q;
            t.block_begin[j] = blocks;
            blocks += (int)nb;
        }
        t.block_begin[t.n] = blocks;
        k_unpack_wgrads_batch<<<blocks, 256, 0, vn_stream(stream)>>>(t);
        VN_LAUNCH_STATUS();
    }
    return VN_OK;
}

extern "C" int vn_nchw_to_rows(const float *src, int32_t B, int32_t C, int64_t S, void *dst, vnDtype dst_dtype,
                               int64_t dst_stride, int64_t lo_off, vnStream stream) {
    VN_CHECK_ARG(src && dst && B > 0 && C > 0 && S > 0 && B <= 65535);
    VN_CHECK_ARG(lo_off >= 0 && (!lo_off || dst_dtype == VN_BF16));
    VN_CHECK_ARG(dst_stride >= C);
    const dim3 grid((unsigned)vn_ceil_div(S, 32), (unsigned)vn_ceil_div(C, 32), (unsigned)B);
    k_nchw_to_rows<<<grid, 256, 0, vn_stream(stream)>>>(src, C, S, dst, (int)dst_dtype, dst_stride, lo_off);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_rows_to_nchw(const void *src, vnDtype src_dtype, int64_t src_stride, int32_t B, int32_t C, int64_t S,
                               float *dst, int32_t sigmoid_first_n, vnStream stream) {
    VN_CHECK_ARG(src && dst && B > 0 && C > 0 && S > 0 && B <= 65535 && src_stride >= C);
    const dim3 grid((unsigned)vn_ceil_div(S, 32), (unsigned)vn_ceil_div(C, 32), (unsigned)B);
    k_rows_to_nchw<<<grid, 256, 0, vn_stream(stream)>>>(src, (int)src_dtype, src_stride, C, S, dst, sigmoid_first_n);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_heads_to_nchw(const float *rows16, int32_t B, int64_t S, float *prob, float *reg, vnStream stream) {
    VN_CHECK_ARG(rows16 && prob && reg && B > 0 && S > 0 && (int64_t)B * S < (1ll << 31));
    k_heads_to_nchw<<<(unsigned)vn_ceil_div((int64_t)B * S, 256), 256, 0, vn_stream(stream)>>>(rows16, B, S, prob, reg);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_cast_rows(const void *src, vnDtype src_dtype, int64_t src_stride, int64_t M, int32_t C, void *dst,
                            vnDtype dst_dtype, int64_t dst_stride, int64_t lo_off, vnStream stream) {
    VN_CHECK_ARG(M >= 0 && C > 0 && (C & 3) == 0 && (src_stride & 3) == 0 && (dst_stride & 3) == 0);
    VN_CHECK_ARG(lo_off >= 0 && (lo_off & 3) == 0);
    if (M == 0) return VN_OK;
    VN_CHECK_ARG(src && dst && (!lo_off || dst_dtype == VN_BF16));
    k_cast_rows<<<gs_blocks(M * (C >> 2)), 256, 0, vn_stream(stream)>>>(src, (int)src_dtype, src_stride, M, C, dst,
                                                                        (int)dst_dtype, dst_stride, lo_off);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" size_t vn_col_sums_workspace_bytes(int64_t M, int32_t C) {
    if (M <= 0 || C <= 0 || (C & 3) || C > 1024) return 0;
    return vn_align(sizeof(float) * (size_t)gs_blocks(M, (256 / (C >> 2)) * 16, 256) * C);
}

extern "C" int vn_col_sums(const void *rows, vnDtype dtype, int64_t stride, int64_t M, int32_t C, float *out,
                           void *workspace, size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(out && M >= 0 && C > 0 && (C & 3) == 0 && C <= 1024 && (stride & 3) == 0);
    if (M == 0) return (int)hipMemsetAsync(out, 0, sizeof(float) * C, vn_stream(stream));
    VN_CHECK_ARG(rows && workspace);
    if (workspace_bytes < vn_col_sums_workspace_bytes(M, C)) return VN_EWORKSPACE;
    const int rpb = 256 / (C >> 2);
    const unsigned blocks = gs_blocks(M, rpb * 16, 256);
    float *part = static_cast<float *>(workspace);
    k_col_sums<<<blocks, 256, 0, vn_stream(stream)>>>(rows, (int)dtype, stride, M, C, part);
    VN_LAUNCH_STATUS();
    k_col_sums_final<<<1, 256, 0, vn_stream(stream)>>>(part, (int)blocks, C, out);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_heads_bwd(const float *d_prob, const float *d_reg, const float *prob, int32_t B, int64_t S,
                            void *d_rows, vnDtype d_dtype, int64_t d_stride, int32_t split, vnStream stream) {
    VN_CHECK_ARG(d_prob && d_reg && prob && d_rows && B > 0 && S > 0 && d_stride >= (split ? 32 : 16));
    VN_CHECK_ARG(d_dtype == VN_BF16 || (d_dtype == VN_F32 && !split));
    k_heads_bwd<<<gs_blocks((int64_t)B * S), 256, 0, vn_stream(stream)>>>(d_prob, d_reg, prob, B, S, d_rows, d_stride,
                                                                          split, d_dtype == VN_F32);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
