// Stand-alone VFELayer.forward(inputs, mask) — /root/reference/voxelnet/model.py:60-82 — forward and backward, for
// callers that compose VFE layers themselves (SURVEY.md §8(b) surface table).  The train path does NOT use this file:
// FeatureLearningNet runs both layers and the voxel max fused (csrc/vfe.hip).  Same semantics as the reference module:
//   h = relu(x W^T + b)                      (K,T,units)          model.py:68-71, 75
//   p = BatchNorm1d(h) over the K*T rows     (train: batch statistics, running stats updated; eval: running stats)
//   agg = max_t p                            (K,1,units)          model.py:77
//   out = concat(p, agg repeated) * mask     (K,T,2*units)        model.py:78-81
// fp32 VALU throughout (a 7->16 / 32->64 MLP: memory-bound, nothing GEMM-shaped for the matrix cores).  One wave per
// voxel, lane = output unit; weights in LDS; every reduction goes through per-workgroup partial rows that are added in
// a fixed order (no atomics: bit-reproducible).
#include "common.h"

namespace {

constexpr int VL_WAVES = 4;

struct VLPlan {
    float *h, *dp, *slab, *slab2, *slabw, *st;   // st: mean | invstd | S | beta | m1 | m2 (6 x units)
    int64_t blocks;
    size_t bytes;
};

VLPlan vl_plan(void *base, int64_t K, int T, int cin, int units) {
    VLPlan p{};
    char *b = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t n) { char *r = b ? b + off : nullptr; off += vn_align(n); return r; };
    int64_t blocks = vn_ceil_div(K > 0 ? K : 1, VL_WAVES * 4);
    if (blocks > 1024) blocks = 1024;
    p.blocks = blocks;
    const size_t M = (size_t)(K > 0 ? K : 1) * T;
    p.h = reinterpret_cast<float *>(take(M * units * sizeof(float)));
    p.dp = reinterpret_cast<float *>(take(M * units * sizeof(float)));
    p.slab = reinterpret_cast<float *>(take((size_t)blocks * 2 * units * sizeof(float)));
    p.slab2 = reinterpret_cast<float *>(take((size_t)blocks * 2 * units * sizeof(float)));
    p.slabw = reinterpret_cast<float *>(take((size_t)blocks * units * (cin + 1) * sizeof(float)));
    p.st = reinterpret_cast<float *>(take((size_t)6 * units * sizeof(float)));
    p.bytes = off;
    return p;
}

// fixed-order combine of the 4 waves' per-lane partials -> one slab row [n_vals][units]
template <int NV>
__device__ __forceinline__ void wg_partials(const float (&v)[NV], int units, float *row, float *red /*[4][NV][64]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) red[(wave * NV + i) * 64 + lane] = v[i];
    __syncthreads();
    if (wave == 0 && lane < units) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            row[i * units + lane] = red[(0 * NV + i) * 64 + lane] + red[(1 * NV + i) * 64 + lane] +
                                    red[(2 * NV + i) * 64 + lane] + red[(3 * NV + i) * 64 + lane];
    }
    __syncthreads();
}

// pass 1: h = relu(x W^T + b) stored; per-workgroup sum / sum of squares of h
__global__ void __launch_bounds__(256) k_vl_p1(const float *__restrict__ x, int64_t K, int T, int cin, int units,
                                               const float *__restrict__ W, const float *__restrict__ bias,
                                               float *__restrict__ h, float *__restrict__ slab) {
    extern __shared__ float sm[];
    float *Wt = sm;                              // [cin][units]
    float *red = sm + cin * units;               // [4][2][64]
    for (int i = threadIdx.x; i < cin * units; i += 256) Wt[(i % cin) * units + i / cin] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float b = lane < units ? bias[lane] : 0.f;
    float s[2] = {0.f, 0.f};
    for (int64_t v = (int64_t)blockIdx.x * VL_WAVES + wave; v < K; v += (int64_t)gridDim.x * VL_WAVES) {
        for (int t = 0; t < T; ++t) {
            const int64_t m = v * T + t;
            const float xr = lane < cin ? x[m * cin + lane] : 0.f;
            float acc = b;
            for (int c = 0; c < cin; ++c) acc = fmaf(__shfl(xr, c, 64), lane < units ? Wt[c * units + lane] : 0.f, acc);
            acc = fmaxf(acc, 0.f);
            if (lane < units) {
                h[m * units + lane] = acc;
                s[0] += acc;
                s[1] += acc * acc;
            }
        }
    }
    wg_partials<2>(s, units, slab + (size_t)blockIdx.x * 2 * units, red);
}

// statistics of the layer: st = mean | invstd | S | beta ; running stats updated in train mode
__global__ void __launch_bounds__(64) k_vl_finalize(const float *__restrict__ slab, int64_t rows, int64_t M, int units,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float *running_mean, float *running_var, int training, float momentum,
                                                    float eps, float *__restrict__ st) {
    const int u = threadIdx.x;
    if (u >= units) return;
    double mean, var;
    if (training) {
        double s1 = 0.0, s2 = 0.0;
        for (int64_t r = 0; r < rows; ++r) { s1 += (double)slab[(r * 2 + 0) * units + u]; s2 += (double)slab[(r * 2 + 1) * units + u]; }
        const double n = (double)M;
        mean = s1 / n;
        var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        running_mean[u] = (float)((1.0 - momentum) * running_mean[u] + momentum * mean);
        running_var[u] = (float)((1.0 - momentum) * running_var[u] + momentum * (n > 1.0 ? var * n / (n - 1.0) : var));
    } else {
        mean = running_mean[u];
        var = running_var[u];
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    st[u] = (float)mean;
    st[units + u] = invstd;
    st[2 * units + u] = gamma[u] * invstd;
    st[3 * units + u] = beta[u];
}

// pass 2: p = S (h - mean) + beta, agg = max_t p, out = [p | agg] * mask
__global__ void __launch_bounds__(256) k_vl_p2(const float *__restrict__ h, const uint8_t *__restrict__ mask, int64_t K, int T,
                                               int units, const float *__restrict__ st, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= units) return;
    const float mean = st[lane], S = st[2 * units + lane], be = st[3 * units + lane];
    for (int64_t v = (int64_t)blockIdx.x * VL_WAVES + wave; v < K; v += (int64_t)gridDim.x * VL_WAVES) {
        float agg = -INFINITY;
        for (int t = 0; t < T; ++t) agg = fmaxf(agg, fmaf(S, h[(v * T + t) * units + lane] - mean, be));
        for (int t = 0; t < T; ++t) {
            const int64_t m = v * T + t;
            const float p = fmaf(S, h[m * units + lane] - mean, be);
            const float mk = mask[m] ? 1.f : 0.f;
            out[m * 2 * units + lane] = p * mk;
            out[m * 2 * units + units + lane] = agg * mk;
        }
    }
}

// backward pass 1: dp = mask*d_out[:, :units] + [t == first argmax_t p] * sum_t mask*d_out[:, units:]; partial sums of
// dp and dp * xhat
__global__ void __launch_bounds__(256) k_vl_b1(const float *__restrict__ h, const uint8_t *__restrict__ mask,
                                               const float *__restrict__ d_out, int64_t K, int T, int units,
                                               const float *__restrict__ st, float *__restrict__ dp, float *__restrict__ slab) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = lane < units;
    const float mean = on ? st[lane] : 0.f, invstd = on ? st[units + lane] : 0.f, S = on ? st[2 * units + lane] : 0.f,
                be = on ? st[3 * units + lane] : 0.f;
    float s[2] = {0.f, 0.f};
    for (int64_t v = (int64_t)blockIdx.x * VL_WAVES + wave; v < K && on; v += (int64_t)gridDim.x * VL_WAVES) {
        float agg = -INFINITY, dagg = 0.f;
        int arg = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t m = v * T + t;
            const float p = fmaf(S, h[m * units + lane] - mean, be);
            if (p > agg) { agg = p; arg = t; }                                   // first maximum (torch.max backward)
            if (mask[m]) dagg += d_out[m * 2 * units + units + lane];
        }
        for (int t = 0; t < T; ++t) {
            const int64_t m = v * T + t;
            float g = mask[m] ? d_out[m * 2 * units + lane] : 0.f;
            if (t == arg) g += dagg;
            dp[m * units + lane] = g;
            s[0] += g;
            s[1] += g * ((h[m * units + lane] - mean) * invstd);
        }
    }
    wg_partials<2>(s, units, slab + (size_t)blockIdx.x * 2 * units, sm);
}

// BatchNorm backward coefficients: st[4u..] = m1 = mean(dp), st[5u..] = m2 = mean(dp * xhat); dgamma, dbeta
__global__ void __launch_bounds__(64) k_vl_bfinal(const float *__restrict__ slab, int64_t rows, int64_t M, int units,
                                                  int training, float *__restrict__ st, float *__restrict__ dgamma,
                                                  float *__restrict__ dbeta) {
    const int u = threadIdx.x;
    if (u >= units) return;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t r = 0; r < rows; ++r) { s1 += (double)slab[(r * 2 + 0) * units + u]; s2 += (double)slab[(r * 2 + 1) * units + u]; }
    dbeta[u] = (float)s1;
    dgamma[u] = (float)s2;
    st[4 * units + u] = training ? (float)(s1 / (double)M) : 0.f;     // eval mode: the statistics are constants
    st[5 * units + u] = training ? (float)(s2 / (double)M) : 0.f;
}

// backward pass 2: dh = S (dp - m1 - xhat m2), dz = dh * (h > 0); dx = dz W; per-workgroup partials of dW | db
template <int CP>
__global__ void __launch_bounds__(256) k_vl_b2(const float *__restrict__ x, const float *__restrict__ h,
                                               const float *__restrict__ dp, int64_t K, int T, int cin, int units,
                                               const float *__restrict__ W, const float *__restrict__ st,
                                               float *__restrict__ dx, float *__restrict__ slabw) {
    extern __shared__ float sm[];
    float *Wu = sm;                               // [units][cin]
    float *red = sm + units * cin;                // [4][64] per pass
    for (int i = threadIdx.x; i < cin * units; i += 256) Wu[i] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = lane < units;
    const float mean = on ? st[lane] : 0.f, invstd = on ? st[units + lane] : 0.f, S = on ? st[2 * units + lane] : 0.f,
                m1 = on ? st[4 * units + lane] : 0.f, m2 = on ? st[5 * units + lane] : 0.f;
    float aw[CP + 1];
#pragma unroll
    for (int c = 0; c <= CP; ++c) aw[c] = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * VL_WAVES + wave; v < K; v += (int64_t)gridDim.x * VL_WAVES) {
        for (int t = 0; t < T; ++t) {
            const int64_t m = v * T + t;
            float dz = 0.f;
            if (on) {
                const float hv = h[m * units + lane];
                const float dh = S * (dp[m * units + lane] - m1 - (hv - mean) * invstd * m2);
                dz = hv > 0.f ? dh : 0.f;
            }
            const float xr = lane < cin ? x[m * cin + lane] : 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < cin) aw[c] = fmaf(dz, __shfl(xr, c, 64), aw[c]);
            aw[CP] += dz;
            if (dx) {
                float a = 0.f;
                for (int u = 0; u < units; ++u) a = fmaf(__shfl(dz, u, 64), lane < cin ? Wu[u * cin + lane] : 0.f, a);
                if (lane < cin) dx[m * cin + lane] = a;
            }
        }
    }
    // slabw[block][units][cin + 1]: combine the four waves in a fixed order, one column at a time
    float *row = slabw + (size_t)blockIdx.x * units * (cin + 1);
#pragma unroll
    for (int c = 0; c <= CP; ++c) {
        if (c < cin || c == CP) {
            red[wave * 64 + lane] = aw[c];
            __syncthreads();
            if (wave == 0 && on) row[lane * (cin + 1) + (c == CP ? cin : c)] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256) k_vl_wreduce(const float *__restrict__ slabw, int64_t rows, int units, int cin,
                                                    float *__restrict__ dW, float *__restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= units * (cin + 1)) return;
    float s = 0.f;
    for (int64_t r = 0; r < rows; ++r) s += slabw[r * units * (cin + 1) + i];
    const int u = i / (cin + 1), c = i - u * (cin + 1);
    if (c == cin) db[u] = s;
    else dW[u * cin + c] = s;
}

bool vl_ok(int64_t K, int T, int cin, int units) {
    return K >= 0 && K < (1ll << 31) / 64 && T > 0 && T <= 4096 && cin > 0 && cin <= 64 && units > 0 && units <= 64;
}

}  // namespace

extern "C" size_t vn_vfe_layer_workspace_bytes(int64_t K, int32_t T, int32_t cin, int32_t units) {
    if (!vl_ok(K, T, cin, units)) return 0;
    return vl_plan(nullptr, K, T, cin, units).bytes;
}

extern "C" int vn_vfe_layer_fwd(const float *inputs, const uint8_t *mask, int64_t K, int32_t T, int32_t cin, int32_t units,
                                const float *weight, const float *bias, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, int32_t training, float momentum, float eps,
                                float *out, void *workspace, size_t workspace_bytes, vnStream stream) {
    if (!vl_ok(K, T, cin, units)) return VN_EUNSUPPORTED;
    VN_CHECK_ARG(weight && bias && gamma && beta && running_mean && running_var && workspace);
    const VLPlan p = vl_plan(workspace, K, T, cin, units);
    if (workspace_bytes < p.bytes) return VN_EWORKSPACE;
    if (K == 0) return VN_OK;
    VN_CHECK_ARG(inputs && mask && out);
    hipStream_t st = vn_stream(stream);
    const size_t lds = ((size_t)cin * units + 4 * 2 * 64) * sizeof(float);
    k_vl_p1<<<(unsigned)p.blocks, 256, lds, st>>>(inputs, K, T, cin, units, weight, bias, p.h, p.slab);
    VN_LAUNCH_STATUS();
    k_vl_finalize<<<1, 64, 0, st>>>(p.slab, p.blocks, K * T, units, gamma, beta, running_mean, running_var, training,
                                    momentum, eps, p.st);
    VN_LAUNCH_STATUS();
    k_vl_p2<<<(unsigned)p.blocks, 256, 0, st>>>(p.h, mask, K, T, units, p.st, out);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// workspace: the forward's (h and the statistics are read back from it).  d_inputs may be NULL.
extern "C" int vn_vfe_layer_bwd(const float *inputs, const uint8_t *mask, const float *d_out, int64_t K, int32_t T,
                                int32_t cin, int32_t units, const float *weight, int32_t training, float *d_inputs,
                                float *d_weight, float *d_bias, float *d_gamma, float *d_beta, void *workspace,
                                size_t workspace_bytes, vnStream stream) {
    if (!vl_ok(K, T, cin, units)) return VN_EUNSUPPORTED;
    VN_CHECK_ARG(weight && d_weight && d_bias && d_gamma && d_beta && workspace);
    const VLPlan p = vl_plan(workspace, K, T, cin, units);
    if (workspace_bytes < p.bytes) return VN_EWORKSPACE;
    hipStream_t st = vn_stream(stream);
    if (K == 0) {
        VN_HIP(hipMemsetAsync(d_weight, 0, sizeof(float) * cin * units, st));
        VN_HIP(hipMemsetAsync(d_bias, 0, sizeof(float) * units, st));
        VN_HIP(hipMemsetAsync(d_gamma, 0, sizeof(float) * units, st));
        VN_HIP(hipMemsetAsync(d_beta, 0, sizeof(float) * units, st));
        return VN_OK;
    }
    VN_CHECK_ARG(inputs && mask && d_out);
    k_vl_b1<<<(unsigned)p.blocks, 256, 4 * 2 * 64 * sizeof(float), st>>>(p.h, mask, d_out, K, T, units, p.st, p.dp, p.slab2);
    VN_LAUNCH_STATUS();
    k_vl_bfinal<<<1, 64, 0, st>>>(p.slab2, p.blocks, K * T, units, training, p.st, d_gamma, d_beta);
    VN_LAUNCH_STATUS();
    const size_t lds = ((size_t)cin * units + 4 * 64) * sizeof(float);
    if (cin <= 8)
        k_vl_b2<8><<<(unsigned)p.blocks, 256, lds, st>>>(inputs, p.h, p.dp, K, T, cin, units, weight, p.st, d_inputs, p.slabw);
    else if (cin <= 32)
        k_vl_b2<32><<<(unsigned)p.blocks, 256, lds, st>>>(inputs, p.h, p.dp, K, T, cin, units, weight, p.st, d_inputs, p.slabw);
    else
        k_vl_b2<64><<<(unsigned)p.blocks, 256, lds, st>>>(inputs, p.h, p.dp, K, T, cin, units, weight, p.st, d_inputs, p.slabw);
    VN_LAUNCH_STATUS();
    k_vl_wreduce<<<(unsigned)vn_ceil_div(units * (cin + 1), 256), 256, 0, st>>>(p.slabw, p.blocks, units, cin, d_weight, d_bias);
    VN_LAUNCH_STATUS();
    return VN_OK;
}
