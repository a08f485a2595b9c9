// Micro-benchmark: what one launch boundary of a dependent chain costs on one stream, by the size and the place of the
// kernel's parameter block.  (a) 16 B of arguments, (b) a 2.2-KB struct by value (the convolution kernels' GGParams), of
// which the kernel reads a few fields and one indexed entry, (c) 16 B of arguments, one a pointer to the same 2.2-KB block
// resident in device memory (written once).  Each launch: `wgs` workgroups of 256 threads, every thread reads one float the
// previous launch wrote and writes one.  Reports us per launch over a chain of `n` launches (hipEvents around the chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big {
    const float *src;
    float *dst;
    int32_t n, which, ticks, pad0;
    int32_t cls[16][23];
    int32_t taps[32][4];
    int32_t pad[20];
};
// every wave stays for `ticks` of the shader clock (s_memtime) after its load has returned: with ~8 us per launch the host runs
// ahead and the time per launch minus the spin is what the GPU itself needs between two dependent launches
__device__ __forceinline__ void spin(long long ticks) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__global__ void __launch_bounds__(256) k_small(const float *src, float *dst, int n, int ticks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float v = i < n ? src[i] : 0.f;
    if (ticks) spin(ticks + (v > 1e30f));
    if (i < n) dst[i] = v + 1.0f;
}
__global__ void __launch_bounds__(256) k_big(const Big p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = p.cls[p.which][3] + p.taps[blockIdx.x & 31][1];
    const float v = i < p.n ? p.src[i] : 0.f;
    if (p.ticks) spin(p.ticks + (v > 1e30f));
    if (i < p.n) p.dst[i] = v + 1.0f + (float)c;
}
__global__ void __launch_bounds__(256) k_ptr(const Big *__restrict__ pp, const float *src, float *dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = pp->cls[pp->which][3] + pp->taps[blockIdx.x & 31][1];
    const float v = i < pp->n ? src[i] : 0.f;
    if (pp->ticks) spin(pp->ticks + (v > 1e30f));
    if (i < pp->n) dst[i] = v + 1.0f + (float)c;
}

int main(int argc, char **argv) {
    const int n_launch = argc > 1 ? atoi(argv[1]) : 4000;
    const int ticks = argc > 2 ? atoi(argv[2]) : 0;     // s_memtime ticks every wave stays (~20000 = 8-10 us)
    for (int wgs : {1, 256, 2048}) {
        const int n = wgs * 256;
        float *a, *b;
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
        CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
        Big h{};
        h.n = n; h.which = 3; h.ticks = ticks;
        Big *d;
        CK(hipMalloc(&d, sizeof(Big)));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < n_launch; ++i) {
                    const float *s = (i & 1) ? b : a;
                    float *t = (i & 1) ? a : b;
                    if (mode == 0) k_small<<<wgs, 256, 0, st>>>(s, t, n, ticks);
                    else if (mode == 1) { h.src = s; h.dst = t; k_big<<<wgs, 256, 0, st>>>(h); }
                    else {
                        if (i == 0) { h.src = nullptr; h.dst = nullptr; CK(hipMemcpyAsync(d, &h, sizeof(Big), hipMemcpyHostToDevice, st)); }
                        k_ptr<<<wgs, 256, 0, st>>>(d, s, t);
                    }
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("wgs %5d spin %4d  %-28s %.2f us per launch (chain of %d, best of 3; sizeof(Big) = %zu)\n", wgs, ticks,
                   mode == 0 ? "16 B of arguments" : mode == 1 ? "2.2-KB struct by value" : "pointer to a resident block", 1e3f * best / n_launch,
                   n_launch, sizeof(Big));
        }
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(d));
    }
    return 0;
}
