// Voxelizer (V1-V4) — replaces /root/reference/voxelnet/utils.py:37-88.
//
// Design (MI355X-first, not a translation of the numpy code):
//   the grid has only D*H*W <= ~1.4 M cells, so instead of sorting points we
//   (1) key:    one thread per point, coalesced float4 load, exact fp32 key math,
//               atomicAdd into a per-cell counter (order-free: a count),
//   (2) scan:   ordered compaction of occupied cells with a 64-bit packed
//               (occupied | count) exclusive scan -> row id (== np.unique's
//               lexicographic z,y,x order) and per-row segment offset,
//   (3) fill:   point indices into their row's segment (arbitrary order),
//   (4) gather: one wave per voxel picks the T smallest point indices in
//               ascending order by repeated wave-min (restores input order, so
//               the result is deterministic although (3) used atomics), loads
//               the points, forms the sequential fp32 centroid sum exactly like
//               numpy, and writes (T,7) features, int64 coords and counts.
// HBM-bound: 16 B/point read, 28*T B/voxel written.  No MFMA (nothing GEMM-shaped).
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

struct VoxWs {
    int32_t *cell;     // [cells] count, then row id (-1 if empty)
    int32_t *key;      // [N]
    unsigned long long *blk;  // [nb + 1]
    int32_t *seg_off;  // [kmax + 1]
    int32_t *cnt;      // [kmax]
    int32_t *lin;      // [kmax]
    int32_t *cursor;   // [kmax]
    int32_t *seg;      // [N]
    size_t bytes;
};

VoxWs carve(void *base, int64_t n, int64_t cells) {
    VoxWs w;
    int64_t kmax = n < cells ? n : cells;
    int64_t nb = vn_ceil_div(cells, SCAN_TILE);
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += vn_align(bytes); return r; };
    w.cell = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * cells));
    w.key = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (n + 1)));
    w.blk = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (nb + 1)));
    w.seg_off = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (kmax + 1)));
    w.cnt = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (kmax + 1)));
    w.lin = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (kmax + 1)));
    w.cursor = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (kmax + 1)));
    w.seg = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (n + 1)));
    w.bytes = off;
    return w;
}

// utils.py:37-61.  float32 add, IEEE-correct float32 divide (hipcc's default for
// '/' on gfx950: v_div_scale/v_div_fmas/v_div_fixup), floor; range test on the
// floored float (NaN fails every compare, like numpy's int cast + bounds).
__device__ __forceinline__ int32_t point_key(float4 p, const vnGrid g) {
    float fz = floorf(__fdiv_rn(__fadd_rn(p.z, g.oz), g.vz));
    float fy = floorf(__fdiv_rn(__fadd_rn(p.y, g.oy), g.vy));
    float fx = floorf(__fdiv_rn(__fadd_rn(p.x, g.ox), g.vx));
    bool ok = (fz >= 0.0f) && (fz < (float)g.D) && (fy >= 0.0f) && (fy < (float)g.H) &&
              (fx >= 0.0f) && (fx < (float)g.W);
    if (!ok) return -1;
    return ((int32_t)fz * g.H + (int32_t)fy) * g.W + (int32_t)fx;
}

__global__ void __launch_bounds__(256) k_key(const float4 *__restrict__ pts, int64_t n, vnGrid g,
                                             int32_t *__restrict__ key, int32_t *__restrict__ cell) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t k = point_key(pts[i], g);
    key[i] = k;
    if (k >= 0) atomicAdd(&cell[k], 1);
}

__device__ __forceinline__ unsigned long long pack(int32_t c) {
    return c > 0 ? ((1ull << 32) | (unsigned long long)(uint32_t)c) : 0ull;
}

// block-wide exclusive scan of one u64 per thread (256 threads); returns the
// exclusive prefix, *total = block sum
__device__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *total) {
    __shared__ unsigned long long wsum[SCAN_THREADS / VN_WAVE];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / VN_WAVE; ++w) {
        unsigned long long s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const int32_t *__restrict__ cell, int64_t cells,
                                                              unsigned long long *__restrict__ blk) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < cells) s += pack(cell[base + j]);
    unsigned long long tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// single block: exclusive scan of the block sums in place; blk[nb] = grand total;
// publishes K and the number of in-range points.
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_blocks(unsigned long long *__restrict__ blk, int64_t nb,
                                                              int32_t *__restrict__ k_out) {
    unsigned long long carry = 0;
    for (int64_t base = 0; base < nb; base += SCAN_THREADS) {
        int64_t i = base + threadIdx.x;
        unsigned long long v = i < nb ? blk[i] : 0ull;
        unsigned long long tot;
        unsigned long long ex = block_excl_scan(v, &tot);
        if (i < nb) blk[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        blk[nb] = carry;
        k_out[0] = (int32_t)(carry >> 32);
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_write(int32_t *__restrict__ cell, int64_t cells,
                                                             const unsigned long long *__restrict__ blk,
                                                             int32_t *__restrict__ seg_off, int32_t *__restrict__ cnt,
                                                             int32_t *__restrict__ lin, int32_t *__restrict__ cursor,
                                                             int64_t nb) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t c[SCAN_ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        c[j] = base + j < cells ? cell[base + j] : 0;
        s += pack(c[j]);
    }
    unsigned long long tot;
    unsigned long long ex = block_excl_scan(s, &tot) + blk[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < cells) {
            if (c[j] > 0) {
                int32_t row = (int32_t)(ex >> 32);
                cell[base + j] = row;
                seg_off[row] = (int32_t)(ex & 0xffffffffull);
                cnt[row] = c[j];
                lin[row] = (int32_t)(base + j);
                cursor[row] = 0;
                ex += pack(c[j]);
            } else {
                cell[base + j] = -1;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long g = blk[nb];
        seg_off[(int32_t)(g >> 32)] = (int32_t)(g & 0xffffffffull);
    }
}

__global__ void __launch_bounds__(256) k_fill(const int32_t *__restrict__ key, int64_t n,
                                              const int32_t *__restrict__ cell, const int32_t *__restrict__ seg_off,
                                              int32_t *__restrict__ cursor, int32_t *__restrict__ seg) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t k = key[i];
    if (k < 0) return;
    int32_t row = cell[k];
    int32_t pos = atomicAdd(&cursor[row], 1);
    seg[seg_off[row] + pos] = (int32_t)i;
}

// one wave per voxel row
__global__ void __launch_bounds__(256) k_gather(const float4 *__restrict__ pts, vnGrid g, int64_t K,
                                                const int32_t *__restrict__ seg_off, const int32_t *__restrict__ cnt,
                                                const int32_t *__restrict__ lin, const int32_t *__restrict__ seg,
                                                int64_t batch_index, int32_t coord_cols, float *__restrict__ feature,
                                                int64_t *__restrict__ coord, int64_t *__restrict__ number,
                                                const int32_t *__restrict__ k_dev) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k_dev != nullptr && row >= (int64_t)k_dev[0]) return;   // K still on the device: capacity launch
    if (row >= K) return;   // wave-uniform
    const int T = g.T;
    const int n = cnt[row];
    const int32_t *s = seg + seg_off[row];
    const int m = n < T ? n : T;

    // T smallest point indices, ascending == first T points in input order
    // (utils.py:78-84).  Repeated wave-min over the segment; n <= 64 keeps the
    // candidate in a register, larger segments re-read (L2-resident) memory.
    int my_idx = -1;     // lane t < m ends up holding the t-th smallest index
    int prev = -1;
    if (n <= 64) {
        const int mine = lane < n ? s[lane] : 0x7fffffff;
        for (int t = 0; t < m; ++t) {
            int cand = mine > prev ? mine : 0x7fffffff;
            prev = vn_wave_min_i32(cand);
            if (lane == t) my_idx = prev;
        }
    } else {
        for (int t = 0; t < m; ++t) {
            int cand = 0x7fffffff;
            for (int j = lane; j < n; j += 64) {
                int v = s[j];
                if (v > prev && v < cand) cand = v;
            }
            prev = vn_wave_min_i32(cand);
            if (lane == t) my_idx = prev;
        }
    }
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < m) p = pts[my_idx];   // utils.py:83 (16-B gather)

    // utils.py:87-88: sequential float32 sum over slots 0..T-1 (numpy reduces
    // axis=1 of a (K,T,3) view slot by slot); padded slots add +0.0f.
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (int t = 0; t < m; ++t) {
        s0 = __fadd_rn(s0, __shfl(p.x, t, 64));
        s1 = __fadd_rn(s1, __shfl(p.y, t, 64));
        s2 = __fadd_rn(s2, __shfl(p.z, t, 64));
    }
    // float32 / int64 -> float64 in numpy; subtraction in float64; store rounds.
    const double dn = (double)m;
    const double c0 = (double)s0 / dn, c1 = (double)s1 / dn, c2 = (double)s2 / dn;
    if (lane < T) {
        float *f = feature + ((int64_t)row * T + lane) * 7;
        f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = p.w;
        f[4] = (float)((double)p.x - c0);
        f[5] = (float)((double)p.y - c1);
        f[6] = (float)((double)p.z - c2);
    }
    if (lane == 0) {
        const int32_t c = lin[row];
        const int64_t z = c / (g.H * g.W), y = (c / g.W) % g.H, x = c % g.W;
        int64_t *o = coord + row * coord_cols;
        if (coord_cols == 4) { o[0] = batch_index; o[1] = z; o[2] = y; o[3] = x; }   // dataset.py:110-117
        else { o[0] = z; o[1] = y; o[2] = x; }
        number[row] = m;   // utils.py:84 (saturates at T)
    }
}

bool grid_ok(const vnGrid *g) {
    return g && g->D > 0 && g->H > 0 && g->W > 0 && g->T > 0 && g->T <= 64 &&
           (int64_t)g->D * g->H * g->W < (1ll << 31) && g->vz > 0 && g->vy > 0 && g->vx > 0;
}

}  // namespace

extern "C" size_t vn_voxelize_workspace_bytes(int64_t n_points, const vnGrid *grid) {
    if (!grid_ok(grid) || n_points < 0) return 0;
    return carve(nullptr, n_points, (int64_t)grid->D * grid->H * grid->W).bytes;
}

extern "C" int vn_voxelize_index(const float *points, int64_t n_points, const vnGrid *grid, void *workspace,
                                 size_t workspace_bytes, int32_t *k_out, vnStream stream) {
    VN_CHECK_ARG(grid_ok(grid) && n_points >= 0 && n_points < (1ll << 31) && workspace && k_out);
    VN_CHECK_ARG(points || n_points == 0);
    const int64_t cells = (int64_t)grid->D * grid->H * grid->W;
    VoxWs w = carve(workspace, n_points, cells);
    if (workspace_bytes < w.bytes) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    VN_HIP(hipMemsetAsync(w.cell, 0, sizeof(int32_t) * cells, st));
    if (n_points > 0) {
        k_key<<<dim3((unsigned)vn_ceil_div(n_points, 256)), dim3(256), 0, st>>>(
            reinterpret_cast<const float4 *>(points), n_points, *grid, w.key, w.cell);
        VN_LAUNCH_STATUS();
    }
    const int64_t nb = vn_ceil_div(cells, SCAN_TILE);
    k_scan_reduce<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st>>>(w.cell, cells, w.blk);
    VN_LAUNCH_STATUS();
    k_scan_blocks<<<dim3(1), dim3(SCAN_THREADS), 0, st>>>(w.blk, nb, k_out);
    VN_LAUNCH_STATUS();
    k_scan_write<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st>>>(w.cell, cells, w.blk, w.seg_off, w.cnt, w.lin,
                                                                     w.cursor, nb);
    VN_LAUNCH_STATUS();
    if (n_points > 0) {
        k_fill<<<dim3((unsigned)vn_ceil_div(n_points, 256)), dim3(256), 0, st>>>(w.key, n_points, w.cell, w.seg_off,
                                                                                 w.cursor, w.seg);
        VN_LAUNCH_STATUS();
    }
    return VN_OK;
}

extern "C" int vn_voxelize_gather(const float *points, int64_t n_points, const vnGrid *grid, void *workspace,
                                  size_t workspace_bytes, int64_t K, int64_t batch_index, int32_t coord_cols,
                                  float *feature, int64_t *coord, int64_t *number, const int32_t *k_dev,
                                  vnStream stream) {
    VN_CHECK_ARG(grid_ok(grid) && n_points >= 0 && workspace && K >= 0 && K <= n_points);
    VN_CHECK_ARG(coord_cols == 3 || coord_cols == 4);
    if (K == 0) return VN_OK;
    VN_CHECK_ARG(points && feature && coord && number);
    const int64_t cells = (int64_t)grid->D * grid->H * grid->W;
    VoxWs w = carve(workspace, n_points, cells);
    if (workspace_bytes < w.bytes) return VN_EWORKSPACE;
    k_gather<<<dim3((unsigned)vn_ceil_div(K, 4)), dim3(256), 0, vn_stream(stream)>>>(
        reinterpret_cast<const float4 *>(points), *grid, K, w.seg_off, w.cnt, w.lin, w.seg, batch_index, coord_cols,
        feature, coord, number, k_dev);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
