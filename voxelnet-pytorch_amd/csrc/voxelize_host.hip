// Host (CPU) entry of the voxelizer — SURVEY.md §8(b) `vn_voxelize_host`.
//
// The reference calls pcl_to_voxels (utils.py:10-100) inside forked DataLoader worker processes (dataset.py:58,
// train.py:77-84, num_workers = 8).  A forked worker cannot use the parent's HIP context, so the literal drop-in of that
// call site needs an entry point that takes HOST pointers and touches no device.  This is it: the same two phases, the
// same arithmetic (float32 add / IEEE divide / floor for the voxel index; first T points in input order; sequential
// float32 centroid sum, float64 divide and subtract as numpy promotes) and the same output formats as
// vn_voxelize_index / vn_voxelize_gather, bit for bit.  The train loop itself does not use it: it voxelizes on the
// device (voxelnet_amd/dataset.py DeviceBatcher), where the (K,T,7) buffers never cross PCIe.
//
// O(N + cells): counting sort on the cell grid (np.unique's z,y,x order falls out of the cell scan), no hashing.
#include "common.h"
#include <math.h>
#include <string.h>

namespace {

struct HostWs {
    int32_t *cell;     // [cells]   point count, then row id (-1 if empty)
    int32_t *key;      // [n]       cell of every point (-1: outside)
    int32_t *seg_off;  // [kmax+1]  first slot of row r in seg
    int32_t *lin;      // [kmax]    cell of row r
    int32_t *seg;      // [n]       point indices grouped by row, input order inside a row
    int64_t k;         // rows found by the index phase (kept at the head of the workspace)
    size_t bytes;
};

HostWs carve(void *base, int64_t n, int64_t cells) {
    HostWs w{};
    const int64_t kmax = n < cells ? n : cells;
    char *p = static_cast<char *>(base);
    size_t off = vn_align(sizeof(int64_t));     // [0]: K of the index phase
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += vn_align(bytes); return r; };
    w.cell = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)cells));
    w.key = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)(n + 1)));
    w.seg_off = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)(kmax + 2)));
    w.lin = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)(kmax + 1)));
    w.seg = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)(n + 1)));
    w.bytes = off;
    return w;
}

bool host_grid_ok(const vnGrid *g) {
    return g && g->D > 0 && g->H > 0 && g->W > 0 && g->T > 0 && g->T <= 64 && (int64_t)g->D * g->H * g->W < (1ll << 31) &&
           g->vz > 0.f && g->vy > 0.f && g->vx > 0.f;
}

// utils.py:37-61 for one point; volatile intermediates keep every operation a rounded float32 one whatever the
// host compiler's contraction / excess-precision settings
inline int32_t host_key(const float *p, const vnGrid &g) {
    volatile float sx = p[0] + g.ox, sy = p[1] + g.oy, sz = p[2] + g.oz;
    volatile float qx = sx / g.vx, qy = sy / g.vy, qz = sz / g.vz;
    const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
    const bool ok = (fz >= 0.0f) && (fz < (float)g.D) && (fy >= 0.0f) && (fy < (float)g.H) && (fx >= 0.0f) && (fx < (float)g.W);
    if (!ok) return -1;      // (NaN fails every compare)
    return ((int32_t)fz * g.H + (int32_t)fy) * g.W + (int32_t)fx;
}

}  // namespace

extern "C" size_t vn_voxelize_host_workspace_bytes(int64_t n_points, const vnGrid *grid) {
    if (n_points < 0 || n_points >= (1ll << 31) - 2 || !host_grid_ok(grid)) return 0;
    return carve(nullptr, n_points, (int64_t)grid->D * grid->H * grid->W).bytes;
}

extern "C" int vn_voxelize_host_index(const float *points, int64_t n_points, const vnGrid *grid, void *workspace,
                                      size_t workspace_bytes, int64_t *k_out) {
    VN_CHECK_ARG(k_out && workspace && n_points >= 0 && n_points < (1ll << 31) - 2 && host_grid_ok(grid) && (points || n_points == 0));
    const int64_t cells = (int64_t)grid->D * grid->H * grid->W;
    HostWs w = carve(workspace, n_points, cells);
    if (workspace_bytes < w.bytes) return VN_EWORKSPACE;
    const vnGrid g = *grid;
    memset(w.cell, 0, sizeof(int32_t) * (size_t)cells);
    for (int64_t i = 0; i < n_points; ++i) {
        const int32_t k = host_key(points + 4 * i, g);
        w.key[i] = k;
        if (k >= 0) ++w.cell[k];
    }
    // occupied cells in ascending cell order == np.unique(voxel_index, axis=0) (utils.py:63)
    int64_t K = 0;
    int32_t off = 0;
    for (int64_t c = 0; c < cells; ++c) {
        const int32_t cnt = w.cell[c];
        if (cnt > 0) {
            w.seg_off[K] = off;
            w.lin[K] = (int32_t)c;
            w.cell[c] = (int32_t)K;
            off += cnt;
            ++K;
        } else {
            w.cell[c] = -1;
        }
    }
    w.seg_off[K] = off;
    // points into their row's segment, in input order (utils.py:78-84 walks the points in order): seg_off[r] is the
    // fill cursor of row r and ends at row r+1's start; shifting the array back by one row restores the offsets
    for (int64_t i = 0; i < n_points; ++i) {
        const int32_t k = w.key[i];
        if (k >= 0) w.seg[w.seg_off[w.cell[k]]++] = (int32_t)i;
    }
    for (int64_t r = K; r > 0; --r) w.seg_off[r] = w.seg_off[r - 1];
    w.seg_off[0] = 0;
    *reinterpret_cast<int64_t *>(workspace) = K;
    *k_out = K;
    return VN_OK;
}

extern "C" int vn_voxelize_host_gather(const float *points, int64_t n_points, const vnGrid *grid, const void *workspace,
                                       size_t workspace_bytes, int64_t K, int64_t batch_index, int32_t coord_cols,
                                       float *feature, int64_t *coord, int64_t *number) {
    VN_CHECK_ARG(workspace && n_points >= 0 && host_grid_ok(grid) && K >= 0 && (coord_cols == 3 || coord_cols == 4));
    VN_CHECK_ARG(K == 0 || (points && feature && coord && number));
    const int64_t cells = (int64_t)grid->D * grid->H * grid->W;
    const HostWs w = carve(const_cast<void *>(workspace), n_points, cells);
    if (workspace_bytes < w.bytes) return VN_EWORKSPACE;
    if (K != *reinterpret_cast<const int64_t *>(workspace)) return VN_EINVAL;     // not the K of the index phase
    const vnGrid g = *grid;
    const int T = g.T;
    for (int64_t r = 0; r < K; ++r) {
        const int32_t *s = w.seg + w.seg_off[r];
        const int n = w.seg_off[r + 1] - w.seg_off[r];
        const int m = n < T ? n : T;
        float *f = feature + r * T * 7;
        // utils.py:87-88: sequential float32 sum over the slots (padded slots add +0), float64 divide and subtract
        volatile float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        for (int t = 0; t < m; ++t) {
            const float *p = points + 4 * (int64_t)s[t];
            f[t * 7 + 0] = p[0]; f[t * 7 + 1] = p[1]; f[t * 7 + 2] = p[2]; f[t * 7 + 3] = p[3];
            s0 = s0 + p[0]; s1 = s1 + p[1]; s2 = s2 + p[2];
        }
        const double dn = (double)m;
        const double c0 = (double)s0 / dn, c1 = (double)s1 / dn, c2 = (double)s2 / dn;
        for (int t = 0; t < T; ++t) {
            float *q = f + t * 7;
            if (t >= m) q[0] = q[1] = q[2] = q[3] = 0.0f;
            q[4] = (float)((double)q[0] - c0);
            q[5] = (float)((double)q[1] - c1);
            q[6] = (float)((double)q[2] - c2);
        }
        const int32_t c = w.lin[r];
        const int64_t z = c / (g.H * g.W), y = (c / g.W) % g.H, x = c % g.W;
        int64_t *o = coord + r * coord_cols;
        if (coord_cols == 4) { o[0] = batch_index; o[1] = z; o[2] = y; o[3] = x; }   // dataset.py:110-117
        else { o[0] = z; o[1] = y; o[2] = x; }
        number[r] = m;                                                              // utils.py:84: saturates at T
    }
    return VN_OK;
}
