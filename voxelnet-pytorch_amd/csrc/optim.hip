// Optimizer tail of the train step (train.py:153-154): clip_grad_norm_(parameters, max_norm) followed by the plain
// SGD update (train.py:130: SGD(lr), no momentum, no weight decay), as TWO launches over a chunk table instead of
// torch's ~12 multi-tensor / elementwise launches over the 104 parameter tensors.
//   total = sqrt(sum_i |g_i|^2);  coef = min(1, max_norm / (total + 1e-6));  g *= coef;  p -= lr * g
// The chunk table (device memory, built once by the caller for a fixed set of tensors) cuts every (param, grad) pair
// into pieces of at most VN_OPT_CHUNK elements; one workgroup per chunk in both kernels.  HBM-bound: 4 B read per
// element in the first pass, 12 B (g read, p read+write; +4 when the scaled gradients are written back) in the second.
// Sums: fp32 per thread -> one fp32 partial per chunk -> every workgroup of the second kernel adds the partials in
// double in the same fixed order (deterministic, no atomics, no third launch).
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_CHUNK = 4096;           // == VN_OPT_CHUNK in the header
static_assert(OPT_CHUNK == VN_OPT_CHUNK, "header and kernel disagree on the chunk size");

__device__ __forceinline__ bool aligned16(const void *a, const void *b) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

__global__ void __launch_bounds__(OPT_THREADS) k_opt_sumsq(const vnParamChunk *__restrict__ chunks, float *__restrict__ partial) {
    const vnParamChunk c = chunks[blockIdx.x];
    const float *__restrict__ g = c.grad;
    float s = 0.f;
    if (aligned16(g, g)) {
        const int n4 = c.n >> 2;
        for (int i = threadIdx.x; i < n4; i += OPT_THREADS) {
            const float4 v = reinterpret_cast<const float4 *>(g)[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += OPT_THREADS) s += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < c.n; i += OPT_THREADS) s += g[i] * g[i];
    }
    __shared__ float red[OPT_THREADS / 64];
    s = vn_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(OPT_THREADS) k_opt_update(const vnParamChunk *__restrict__ chunks, int n_chunks,
                                                           const float *__restrict__ partial, float max_norm, float lr,
                                                           int scale_grads, float *__restrict__ total_norm) {
    // every workgroup recomputes the (same) total from the partials: n_chunks * 4 B from L2
    double s = 0.0;
    for (int i = threadIdx.x; i < n_chunks; i += OPT_THREADS) s += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double red[OPT_THREADS / 64];
    __shared__ float coef_s;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
        const float total = (float)sqrt(t);
        const float coef = max_norm / (total + 1e-6f);         // torch.nn.utils.clip_grad_norm_
        coef_s = coef < 1.f ? coef : 1.f;                        // clamp(max=1.0); NaN stays NaN as in torch
        if (coef != coef) coef_s = coef;
        if (blockIdx.x == 0 && total_norm) *total_norm = total;
    }
    __syncthreads();
    const float coef = coef_s;
    const vnParamChunk c = chunks[blockIdx.x];
    float *__restrict__ p = c.param;
    float *__restrict__ g = c.grad;
    if (aligned16(p, g)) {
        const int n4 = c.n >> 2;
        for (int i = threadIdx.x; i < n4; i += OPT_THREADS) {
            float4 gv = reinterpret_cast<const float4 *>(g)[i];
            float4 pv = reinterpret_cast<float4 *>(p)[i];
            gv.x *= coef; gv.y *= coef; gv.z *= coef; gv.w *= coef;
            pv.x -= lr * gv.x; pv.y -= lr * gv.y; pv.z -= lr * gv.z; pv.w -= lr * gv.w;
            reinterpret_cast<float4 *>(p)[i] = pv;
            if (scale_grads) reinterpret_cast<float4 *>(g)[i] = gv;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < c.n; i += OPT_THREADS) {
            const float gv = g[i] * coef;
            p[i] -= lr * gv;
            if (scale_grads) g[i] = gv;
        }
    } else {
        for (int i = threadIdx.x; i < c.n; i += OPT_THREADS) {
            const float gv = g[i] * coef;
            p[i] -= lr * gv;
            if (scale_grads) g[i] = gv;
        }
    }
}

}  // namespace

extern "C" size_t vn_clip_sgd_workspace_bytes(int32_t n_chunks) {
    if (n_chunks <= 0) return 0;
    return vn_align(sizeof(float) * (size_t)n_chunks);
}

extern "C" int vn_clip_sgd(const vnParamChunk *chunks, int32_t n_chunks, float max_norm, float lr, int32_t scale_grads,
                           void *workspace, size_t workspace_bytes, float *total_norm, vnStream stream) {
    VN_CHECK_ARG(chunks && n_chunks > 0 && workspace && max_norm > 0.f);
    if (workspace_bytes < vn_clip_sgd_workspace_bytes(n_chunks)) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    float *partial = static_cast<float *>(workspace);
    k_opt_sumsq<<<n_chunks, OPT_THREADS, 0, st>>>(chunks, partial);
    VN_LAUNCH_STATUS();
    k_opt_update<<<n_chunks, OPT_THREADS, 0, st>>>(chunks, n_chunks, partial, max_norm, lr, scale_grads, total_norm);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
