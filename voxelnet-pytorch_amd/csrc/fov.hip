// Camera field-of-view crop of a raw Velodyne sweep — the reference's offline preprocessing step
// (voxelnet/preprocess_data.py:42-103: prepare_velo_points, project_velo_to_img, the in-image test of align_img_and_velo;
// main() rewrites every .bin with the surviving [x, y, z, reflectance] rows, :151-154).  It decides N (121k -> ~20-24k
// points per frame) in front of the voxelizer, so on raw drives it belongs to the input pipeline.
//
// A point survives when, in float32 like the reference's float32 matrices,
//   reflectance > 0                                               (preprocess_data.py:45)
//   c = R_rect . (Tr_velo_to_cam . [x, y, z, 1]) has c_z >= 0      (:54-56)
//   (u, v, w) = P . c;  col = rint(u / w), row = rint(v / w)       (:57, 59, 83-84: np.round = round half to even)
//   0 < col < cols and 0 < row < rows                              (:86)
// and the survivors keep their input order.  Every 4-term product sum is evaluated as the fused chain
// fma(m3, p3, fma(m2, p2, fma(m1, p1, m0 * p0))) — the order and contraction of a k-sequential sgemm micro-kernel; a point
// changes sides only if a coordinate lies within an ulp of a decision boundary (tests/test_gpu_fov.py pins the index set
// to a fixture made by the imported reference functions on a bundled KITTI frame).
// Three launches: flags + per-workgroup counts, one-workgroup scan of the counts, order-preserving compaction.
#include "common.h"

namespace {

struct FovCalib {
    float P[12], T[12], R[9];     // P2 (3x4), Tr_velo_to_cam (top 3 rows of the 4x4), R0_rect (3x3 of the 4x4)
};

__device__ __forceinline__ float dot4(const float *m, float a, float b, float c, float d) {
    return fmaf(m[3], d, fmaf(m[2], c, fmaf(m[1], b, m[0] * a)));
}

__device__ __forceinline__ bool fov_keep(const float4 p, const FovCalib &c, int rows, int cols) {
    if (!(p.w > 0.f)) return false;
    // T is 4x4 with last row (0,0,0,1): t3 = 1; R likewise: its fourth column / row contribute 0 * t + ... exactly
    const float t0 = dot4(c.T + 0, p.x, p.y, p.z, 1.f), t1 = dot4(c.T + 4, p.x, p.y, p.z, 1.f), t2 = dot4(c.T + 8, p.x, p.y, p.z, 1.f);
    const float r0[4] = {c.R[0], c.R[1], c.R[2], 0.f}, r1[4] = {c.R[3], c.R[4], c.R[5], 0.f}, r2[4] = {c.R[6], c.R[7], c.R[8], 0.f};
    const float c0 = dot4(r0, t0, t1, t2, 1.f), c1 = dot4(r1, t0, t1, t2, 1.f), c2 = dot4(r2, t0, t1, t2, 1.f);
    if (!(c2 >= 0.f)) return false;
    const float u = dot4(c.P + 0, c0, c1, c2, 1.f), v = dot4(c.P + 4, c0, c1, c2, 1.f), w = dot4(c.P + 8, c0, c1, c2, 1.f);
    const float col = rintf(u / w), row = rintf(v / w);      // (IEEE divide: -fhip-fp32-correctly-rounded-divide-sqrt)
    return col < (float)cols && row < (float)rows && row > 0.f && col > 0.f;   // NaN (w == 0) fails every comparison
}

__global__ void __launch_bounds__(256) k_fov_flags(const float4 *__restrict__ pts, int64_t n, FovCalib c, int rows, int cols,
                                                   uint8_t *__restrict__ flags, int32_t *__restrict__ block_counts) {
    __shared__ int wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n && fov_keep(pts[i], c, rows, cols);
    if (i < n) flags[i] = keep ? 1 : 0;
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts in place (one workgroup; nb <= a few thousand), total -> *count
__global__ void __launch_bounds__(1024) k_fov_scan(int32_t *__restrict__ block_counts, int nb, int32_t *__restrict__ count) {
    __shared__ int part[1024];
    const int per = (nb + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int s = 0;
    for (int j = 0; j < per; ++j)
        if (b0 + j < nb) s += block_counts[b0 + j];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int j = 0; j < per; ++j)
        if (b0 + j < nb) {
            const int v = block_counts[b0 + j];
            block_counts[b0 + j] = run;
            run += v;
        }
    if (threadIdx.x == 1023) *count = part[1023];
}

__global__ void __launch_bounds__(256) k_fov_compact(const float4 *__restrict__ pts, int64_t n, const uint8_t *__restrict__ flags,
                                                     const int32_t *__restrict__ block_offsets, float4 *__restrict__ out,
                                                     int32_t *__restrict__ index, const int32_t *__restrict__ count) {
    __shared__ int wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n && flags[i];
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int base = block_offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (keep) {
        const int dst = base + __popcll(m & ((1ull << lane) - 1ull));
        out[dst] = pts[i];
        if (index) index[dst] = (int32_t)i;
    }
    // rows past the count: NaN points.  Every range test downstream drops them (the voxelizer's key compares fail on a
    // NaN), so a consumer may run on the whole capacity-sized buffer without reading the count back to the host.
    // (Row i >= count is written by nobody else: survivors only land below the count.)
    if (i < n && i >= (int64_t)count[0]) {
        const float q = __builtin_nanf("");
        out[i] = make_float4(q, q, q, q);
    }
}

}  // namespace

extern "C" size_t vn_fov_crop_workspace_bytes(int64_t n) {
    if (n < 0 || n >= (1ll << 31)) return 0;
    return vn_align((size_t)n) + vn_align((size_t)vn_ceil_div(n > 0 ? n : 1, 256) * sizeof(int32_t));
}

extern "C" int vn_fov_crop(const float *points, int64_t n, const float *P_3x4, const float *Tr_velo_to_cam_4x4,
                           const float *R_rect_4x4, int32_t image_rows, int32_t image_cols, float *out_points,
                           int32_t *out_index, int32_t *out_count, void *workspace, size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(n >= 0 && n < (1ll << 31) && P_3x4 && Tr_velo_to_cam_4x4 && R_rect_4x4 && out_count && image_rows > 0 && image_cols > 0);
    VN_CHECK_ARG(n == 0 || (points && out_points && workspace));
    if (workspace_bytes < vn_fov_crop_workspace_bytes(n)) return VN_EWORKSPACE;
    if ((reinterpret_cast<uintptr_t>(points) & 15) || (reinterpret_cast<uintptr_t>(out_points) & 15)) return VN_EUNSUPPORTED;
    hipStream_t st = vn_stream(stream);
    if (n == 0) {
        VN_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
        return VN_OK;
    }
    FovCalib c;
    for (int i = 0; i < 12; ++i) { c.P[i] = P_3x4[i]; c.T[i] = Tr_velo_to_cam_4x4[i]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.R[i * 3 + j] = R_rect_4x4[i * 4 + j];
    uint8_t *flags = static_cast<uint8_t *>(workspace);
    int32_t *counts = reinterpret_cast<int32_t *>(static_cast<char *>(workspace) + vn_align((size_t)n));
    const int nb = (int)vn_ceil_div(n, 256);
    k_fov_flags<<<nb, 256, 0, st>>>(reinterpret_cast<const float4 *>(points), n, c, image_rows, image_cols, flags, counts);
    VN_LAUNCH_STATUS();
    k_fov_scan<<<1, 1024, 0, st>>>(counts, nb, out_count);
    VN_LAUNCH_STATUS();
    k_fov_compact<<<nb, 256, 0, st>>>(reinterpret_cast<const float4 *>(points), n, flags, counts,
                                      reinterpret_cast<float4 *>(out_points), out_index, out_count);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
