// Micro-benchmark: cost of accumulating per-workgroup BatchNorm partial sums into 2*C shared int64 counters at the end
// of a kernel, instead of writing a slab row per workgroup and running a finalize launch.
//   MODE 0: slab row store (baseline)      MODE 1: agent-scope 64-bit atomic adds on 2*C addresses
//   MODE 2: one replica of the counters per XCD (HW_REG_XCC_ID), workgroup... agent-scope atomics on it
//   MODE 3: replica per XCD, atomics without the system-coherence bits (executed in that XCD's L2)
// Integer (fixed-point) adds commute exactly, so the totals are bit-reproducible whatever the arrival order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const float *__restrict__ src, long long *__restrict__ cnt, float *__restrict__ slab, int work,
                                         int *__restrict__ xcd_seen) {
    // a little streaming work so that workgroups do not all arrive in the same cycle
    float s = 0.f;
    for (int i = 0; i < work; ++i) s += src[((size_t)blockIdx.x * work + i) * 256 + threadIdx.x];
    const long long v = (long long)__float2ll_rn(s * 1048576.0f) | 1;
    if (MODE == 0) {
        slab[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    } else if (MODE == 1) {
        __hip_atomic_fetch_add(cnt + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {
        const int x = xcc_id();
        __hip_atomic_fetch_add(cnt + x * 256 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const int x = xcc_id();
        if (threadIdx.x == 0) xcd_seen[blockIdx.x] = x;
        __hip_atomic_fetch_add(cnt + x * 256 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ void k_sum(const long long *cnt, int reps, long long *out) {
    long long t = 0;
    for (int r = 0; r < reps; ++r) t += cnt[r * 256 + threadIdx.x];
    out[threadIdx.x] = t;
}

template <int MODE>
void run(int W, int work, const float *src, long long *cnt, float *slab, int *seen, long long *out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = MODE >= 2 ? 8 : 1;
    for (int i = 0; i < 3; ++i) k<MODE><<<W, 256>>>(src, cnt, slab, work, seen);
    CK(hipMemset(cnt, 0, 8 * 256 * 8));
    CK(hipDeviceSynchronize());
    k<MODE><<<W, 256>>>(src, cnt, slab, work, seen);
    k_sum<<<1, 256>>>(cnt, reps, out);
    CK(hipDeviceSynchronize());
    std::vector<long long> h1(256), h2(256);
    CK(hipMemcpy(h1.data(), out, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(cnt, 0, 8 * 256 * 8));
    k<MODE><<<W, 256>>>(src, cnt, slab, work, seen);
    k_sum<<<1, 256>>>(cnt, reps, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h2.data(), out, 256 * 8, hipMemcpyDeviceToHost));
    bool same = true;
    for (int i = 0; i < 256; ++i) same &= h1[i] == h2[i];
    const int N = 50;
    CK(hipEventRecord(a));
    for (int i = 0; i < N; ++i) k<MODE><<<W, 256>>>(src, cnt, slab, work, seen);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("  mode %d  W %5d  work %3d : %7.2f us/launch   totals %s (cnt[0] = %lld)\n", MODE, W, work, 1e3 * ms / N,
           MODE == 0 ? "n/a" : (same ? "reproducible" : "DIFFER"), h1[0]);
    if (MODE == 3) {
        std::vector<int> hs(W);
        CK(hipMemcpy(hs.data(), seen, W * 4, hipMemcpyDeviceToHost));
        int hist[16] = {0}, rr = 0;
        for (int i = 0; i < W; ++i) { hist[hs[i] & 15]++; rr += (hs[i] == (i & 7)); }
        printf("    XCC_ID histogram:");
        for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
        printf("   (workgroups with XCC_ID == blockIdx %% 8: %d of %d)\n", rr, W);
    }
}

int main() {
    const int Wmax = 4480, workmax = 16;
    float *src, *slab;
    long long *cnt, *out;
    int *seen;
    CK(hipMalloc(&src, (size_t)Wmax * workmax * 256 * 4));
    CK(hipMemset(src, 0, (size_t)Wmax * workmax * 256 * 4));
    {
        std::vector<float> h((size_t)Wmax * workmax * 256);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
        CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&slab, (size_t)Wmax * 256 * 4));
    CK(hipMalloc(&cnt, 8 * 256 * 8));
    CK(hipMalloc(&out, 256 * 8));
    CK(hipMalloc(&seen, Wmax * 4));
    const int Ws[] = {156, 300, 440, 1100, 2048, 4422};
    for (int work : {1, 16})
        for (int W : Ws) {
            run<0>(W, work, src, cnt, slab, seen, out);
            run<1>(W, work, src, cnt, slab, seen, out);
            run<2>(W, work, src, cnt, slab, seen, out);
            run<3>(W, work, src, cnt, slab, seen, out);
        }
    return 0;
}
