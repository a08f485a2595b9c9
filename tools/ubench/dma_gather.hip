// Micro-benchmark: how fast can 4-wave workgroups stage [rows x 128 B] pieces into LDS by LDS-DMA, as a function of
// the row stride in memory (128 B = contiguous rows, 256/512 B = half / quarter of a wider row per K step), the
// table size (L2 / Infinity Cache) and the number of stages in flight?   usage: dma_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PIECES, int NS>
__global__ void __launch_bounds__(256) k_dma(const char *tab, uint32_t tab_bytes, int stride, int iters, int rows_total, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    constexpr int STAGE = PIECES * 4 * 1024;
    constexpr int AHEAD = (NS - 2) * PIECES;
    uint32_t row0 = (uint32_t)blockIdx.x * 977u;
    auto stage = [&](int it, int buf) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            uint32_t r = (row0 + (uint32_t)it * (PIECES * 32) + (p * 4 + wave) * 8 + (lane >> 3)) % (uint32_t)rows_total;
            uint32_t off = r * (uint32_t)stride + (lane & 7) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + buf * STAGE + (p * 4 + wave) * 1024), 16, off, 0, 0, 0);
        }
    };
    for (int j = 0; j < NS - 1; ++j) stage(j, j);
    float acc = 0.f;
    int buf = 0, nbuf = NS - 1;
    for (int it = 0; it < iters; ++it) {
        if (NS > 2 && it + NS - 2 < iters) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + NS - 1 < iters) stage(it + NS - 1, nbuf);
        acc += *reinterpret_cast<const float *>(smem + buf * STAGE + threadIdx.x * 16);
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int PIECES, int NS>
void run(const char *tab, size_t tab_bytes, int stride, int blocks_per_cu, float *sink) {
    const int iters = 400;
    const size_t lds = (size_t)NS * PIECES * 4096;
    CK(hipFuncSetAttribute((const void *)&k_dma<PIECES, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int rows_total = (int)(tab_bytes / stride);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_dma<PIECES, NS><<<grid, 256, lds>>>(tab, (uint32_t)tab_bytes, stride, iters, rows_total, sink);
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k_dma<PIECES, NS><<<grid, 256, lds>>>(tab, (uint32_t)tab_bytes, stride, iters, rows_total, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = (double)grid * iters * PIECES * 4096;
    printf("pieces/wave %2d stages %d blocks/CU %d stride %4d table %6.1f MB : %7.2f TB/s  (%5.1f B/clk/CU @2.4GHz)\n", PIECES, NS,
           blocks_per_cu, stride, tab_bytes / 1048576.0, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
}

int main() {
    char *tab; float *sink;
    const size_t cap = 512u << 20;
    CK(hipMalloc(&tab, cap)); CK(hipMemset(tab, 1, cap)); CK(hipMalloc(&sink, 4));
    for (size_t mb : {8, 128}) {
        for (int stride : {128, 256, 512, 1536}) {
            run<8, 2>(tab, mb << 20, stride, 2, sink);
            run<8, 4>(tab, mb << 20, stride, 1, sink);
            run<4, 4>(tab, mb << 20, stride, 2, sink);
            run<2, 8>(tab, mb << 20, stride, 4, sink);
        }
    }
    return 0;
}
