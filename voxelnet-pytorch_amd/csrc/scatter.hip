// Sparse -> dense scatter (S1) — replaces model.py:102-106
// (torch.sparse.FloatTensor(coords.t(), voxelwise, [B,D,H,W,128]).to_dense()).
//
// HBM-bound: the zero-fill of the (B,D,H,W,C) grid dominates (721 MB fp32 /
// 360 MB bf16 per car sample); the K occupied rows are 512 B (fp32) each and are
// written as 16-B (fp32) / 8-B (bf16) stores by consecutive lanes, coalesced.
// Deterministic: coordinates are unique per sample, rows never collide.
#include "common.h"

namespace {

__device__ __forceinline__ int64_t site_of(const int64_t *c, int B, int D, int H, int W) {
    const int64_t b = c[0], z = c[1], y = c[2], x = c[3];
    if (b < 0 || b >= B || z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return -1;
    return ((b * D + z) * H + y) * W + x;
}

// 16-B zero fill, grid-stride
__global__ void __launch_bounds__(256) k_zero(uint4 *__restrict__ p, int64_t n16) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = z;
}

// one thread per (row, 4-channel group)
template <int MODE>  // 0: f32, 1: bf16, 2: bf16 split (hi|lo)
__global__ void __launch_bounds__(256) k_scatter(const float *__restrict__ vw, const int64_t *__restrict__ coord,
                                                 int64_t K, int C, int B, int D, int H, int W, void *__restrict__ dense,
                                                 int dense_channels) {
    const int groups = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * groups) return;
    const int64_t k = i / groups;
    const int cg = (int)(i - k * groups) << 2;
    const int64_t site = site_of(coord + 4 * k, B, D, H, W);
    if (site < 0) return;
    const float4 v = vw ? *reinterpret_cast<const float4 *>(vw + k * C + cg) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0) {
        *reinterpret_cast<float4 *>(static_cast<float *>(dense) + site * dense_channels + cg) = v;
    } else {
        bf16_t *o = static_cast<bf16_t *>(dense) + site * dense_channels + cg;
        bf16x4_t hi, lo;
        bf16_t h, l;
        vn_split_bf16(v.x, h, l); hi[0] = h; lo[0] = l;
        vn_split_bf16(v.y, h, l); hi[1] = h; lo[1] = l;
        vn_split_bf16(v.z, h, l); hi[2] = h; lo[2] = l;
        vn_split_bf16(v.w, h, l); hi[3] = h; lo[3] = l;
        *reinterpret_cast<bf16x4_t *>(o) = hi;
        if (MODE == 2) *reinterpret_cast<bf16x4_t *>(o + C) = lo;
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256) k_gather_rows(const void *__restrict__ dd, const int64_t *__restrict__ coord,
                                                     int64_t K, int C, int B, int D, int H, int W,
                                                     float *__restrict__ dvw) {
    const int groups = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * groups) return;
    const int64_t k = i / groups;
    const int cg = (int)(i - k * groups) << 2;
    const int64_t site = site_of(coord + 4 * k, B, D, H, W);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (site >= 0) {
        if (BF16) {
            bf16x4_t t = *reinterpret_cast<const bf16x4_t *>(static_cast<const bf16_t *>(dd) + site * C + cg);
            v = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        } else {
            v = *reinterpret_cast<const float4 *>(static_cast<const float *>(dd) + site * C + cg);
        }
    }
    *reinterpret_cast<float4 *>(dvw + k * C + cg) = v;
}

}  // namespace

extern "C" int vn_scatter_dense_fwd(const float *voxelwise, const int64_t *coord, int64_t K, int32_t C, int32_t B,
                                    int32_t D, int32_t H, int32_t W, void *dense, vnDtype dense_dtype,
                                    int32_t dense_channels, int32_t split, vnStream stream) {
    VN_CHECK_ARG(dense && K >= 0 && C > 0 && (C & 3) == 0 && B > 0 && D > 0 && H > 0 && W > 0);
    VN_CHECK_ARG(K == 0 || (voxelwise && coord));
    VN_CHECK_ARG(dense_channels == (split ? 2 * C : C));
    VN_CHECK_ARG(!split || dense_dtype == VN_BF16);
    hipStream_t st = vn_stream(stream);
    const int64_t sites = (int64_t)B * D * H * W;
    const int64_t bytes = sites * dense_channels * (dense_dtype == VN_BF16 ? 2 : 4);
    VN_CHECK_ARG((bytes & 15) == 0);
    k_zero<<<dim3(256 * 8), dim3(256), 0, st>>>(static_cast<uint4 *>(dense), bytes >> 4);
    VN_LAUNCH_STATUS();
    if (K == 0) return VN_OK;
    const unsigned blocks = (unsigned)vn_ceil_div(K * (C >> 2), 256);
    if (dense_dtype == VN_F32)
        k_scatter<0><<<blocks, 256, 0, st>>>(voxelwise, coord, K, C, B, D, H, W, dense, dense_channels);
    else if (!split)
        k_scatter<1><<<blocks, 256, 0, st>>>(voxelwise, coord, K, C, B, D, H, W, dense, dense_channels);
    else
        k_scatter<2><<<blocks, 256, 0, st>>>(voxelwise, coord, K, C, B, D, H, W, dense, dense_channels);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_scatter_dense_update(const float *voxelwise, const int64_t *coord, int64_t K, int32_t C, int32_t B,
                                       int32_t D, int32_t H, int32_t W, void *dense, vnDtype dense_dtype,
                                       int32_t dense_channels, int32_t split, vnStream stream) {
    VN_CHECK_ARG(dense && K >= 0 && C > 0 && (C & 3) == 0 && B > 0 && D > 0 && H > 0 && W > 0);
    VN_CHECK_ARG(K == 0 || coord);
    VN_CHECK_ARG(dense_channels == (split ? 2 * C : C));
    VN_CHECK_ARG(!split || dense_dtype == VN_BF16);
    if (K == 0) return VN_OK;
    hipStream_t st = vn_stream(stream);
    const unsigned blocks = (unsigned)vn_ceil_div(K * (C >> 2), 256);
    if (dense_dtype == VN_F32)
        k_scatter<0><<<blocks, 256, 0, st>>>(voxelwise, coord, K, C, B, D, H, W, dense, dense_channels);
    else if (!split)
        k_scatter<1><<<blocks, 256, 0, st>>>(voxelwise, coord, K, C, B, D, H, W, dense, dense_channels);
    else
        k_scatter<2><<<blocks, 256, 0, st>>>(voxelwise, coord, K, C, B, D, H, W, dense, dense_channels);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

extern "C" int vn_scatter_dense_bwd(const void *d_dense, vnDtype dtype, const int64_t *coord, int64_t K, int32_t C,
                                    int32_t B, int32_t D, int32_t H, int32_t W, float *d_voxelwise, vnStream stream) {
    VN_CHECK_ARG(K >= 0 && C > 0 && (C & 3) == 0 && B > 0 && D > 0 && H > 0 && W > 0);
    if (K == 0) return VN_OK;
    VN_CHECK_ARG(d_dense && coord && d_voxelwise);
    const unsigned blocks = (unsigned)vn_ceil_div(K * (C >> 2), 256);
    if (dtype == VN_BF16)
        k_gather_rows<true><<<blocks, 256, 0, vn_stream(stream)>>>(d_dense, coord, K, C, B, D, H, W, d_voxelwise);
    else
        k_gather_rows<false><<<blocks, 256, 0, vn_stream(stream)>>>(d_dense, coord, K, C, B, D, H, W, d_voxelwise);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
