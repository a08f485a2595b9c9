// Micro-benchmark: cost of staging a [rows x 128 B] stage into LDS per step, LDS-DMA vs register staging
// (global_load_dwordx4 + ds_write_b128), with MFMAs and fragment reads beside it like the conv loop.
// Reports clk per step per workgroup (s_memtime on wave 0) for 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: LDS-DMA pieces in a burst; 1: LDS-DMA interleaved with MFMA groups; 2: register staging (loads issued a
// step ahead, ds_write after the barrier); NM = MFMAs per wave per step (0: none)
template <int PIECES, int MODE, int NM>
__device__ __forceinline__ void body(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    constexpr int STAGE = PIECES * 4 * 1024;
    const uint32_t row0 = (uint32_t)blockIdx.x * 977u;
    auto off_of = [&](int it, int p) {
        uint32_t r = (row0 + (uint32_t)it * (PIECES * 32) + (p * 4 + wave) * 8 + (lane >> 3)) % (uint32_t)rows_total;
        return r * 256u + (lane & 7) * 16;
    };
    f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float4 regs[PIECES];
    if (MODE == 2) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) regs[p] = *reinterpret_cast<const float4 *>(tab + off_of(0, p));
    } else {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + (p * 4 + wave) * 1024), 16, off_of(0, p), 0, 0, 0);
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (MODE == 2) {
            // registers hold stage `it`: write them to LDS, then start the loads of stage it+1
            __builtin_amdgcn_s_barrier();      // readers of buf from two steps ago are done
#pragma unroll
            for (int p = 0; p < PIECES; ++p)
                *reinterpret_cast<float4 *>(smem + buf * STAGE + (p * 4 + wave) * 1024 + lane * 16) = regs[p];
#pragma unroll
            for (int p = 0; p < PIECES; ++p) regs[p] = *reinterpret_cast<const float4 *>(tab + off_of(it + 1, p));
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (MODE == 0) {
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + (buf ^ 1) * STAGE + (p * 4 + wave) * 1024), 16,
                                                             off_of(it + 1, p), 0, 0, 0);
            }
        }
        // fragment reads + MFMAs on stage buf
        const char *l = smem + buf * STAGE + (lane & 15) * 128 + (lane >> 4) * 16;
#pragma unroll
        for (int g = 0; g < (NM > 0 ? NM / 4 : 0); ++g) {
            bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(l + (g & 7) * 2048);
            bf16x8_t b = *reinterpret_cast<const bf16x8_t *>(l + ((g + 3) & 7) * 2048 + 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
            if (MODE == 1) {
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    if (p * (NM / 4) / PIECES == g)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(smem + (buf ^ 1) * STAGE + (p * 4 + wave) * 1024), 16,
                                                                 off_of(it + 1, p), 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (MODE == 2) s += regs[0].x;
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_8_0_0(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,0,0>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_8_2_0(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,2,0>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_8_0_32(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,0,32>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_8_1_32(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,1,32>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_8_2_32(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,2,32>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_4_0_32(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<4,0,32>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_4_2_32(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<4,2,32>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_8_0_64(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,0,64>(tab, tab_bytes, iters, rows_total, sink, clk); }
__global__ void __launch_bounds__(256) k_8_2_64(const char *tab, uint32_t tab_bytes, int iters, int rows_total, float *sink, long long *clk) { body<8,2,64>(tab, tab_bytes, iters, rows_total, sink, clk); }

typedef void (*kern_t)(const char *, uint32_t, int, int, float *, long long *);
void run(kern_t kf, int PIECES, int MODE, int NM, const char *tab, size_t tab_bytes, int bpc, float *sink, long long *clk) {
    const int iters = 200;
    const size_t lds = 2 * (size_t)PIECES * 4096;
    CK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256 * bpc;
    const int rows_total = (int)(tab_bytes / 256) - PIECES * 32 * 2;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    kf<<<grid, 256, lds>>>(tab, (uint32_t)tab_bytes, iters, rows_total, sink, clk);
    CK(hipEventRecord(a));
    kf<<<grid, 256, lds>>>(tab, (uint32_t)tab_bytes, iters, rows_total, sink, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    long long h[8]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const char *names[3] = {"LDS-DMA burst      ", "LDS-DMA interleaved", "register staging   "};
    printf("%s pieces/wave %d MFMA/wave/step %3d WG/CU %d: %6.0f clk/step (s_memtime wave0 of WG0)  wall %.1f us -> %.0f clk/step @2.4GHz\n",
           names[MODE], PIECES, NM, bpc, (double)h[0] / iters, ms * 1e3, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    char *tab; float *sink; long long *clk;
    const size_t cap = 8u << 20;     // L2-resident
    CK(hipMalloc(&tab, cap)); CK(hipMemset(tab, 1, cap)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&clk, 8 * 1024));
    for (int bpc : {1, 2}) {
        run(k_8_0_0, 8, 0, 0, tab, cap, bpc, sink, clk);
        run(k_8_2_0, 8, 2, 0, tab, cap, bpc, sink, clk);
        run(k_8_0_32, 8, 0, 32, tab, cap, bpc, sink, clk);
        run(k_8_1_32, 8, 1, 32, tab, cap, bpc, sink, clk);
        run(k_8_2_32, 8, 2, 32, tab, cap, bpc, sink, clk);
        run(k_4_0_32, 4, 0, 32, tab, cap, bpc, sink, clk);
        run(k_4_2_32, 4, 2, 32, tab, cap, bpc, sink, clk);
        run(k_8_0_64, 8, 0, 64, tab, cap, bpc, sink, clk);
        run(k_8_2_64, 8, 2, 64, tab, cap, bpc, sink, clk);
    }
    return 0;
}
