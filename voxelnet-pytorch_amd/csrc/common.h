// Shared helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/voxelnet_hip.h"

#define VN_WAVE 64

#define VN_CHECK_ARG(cond) do { if (!(cond)) return VN_EINVAL; } while (0)
#define VN_LAUNCH_STATUS() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
#define VN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; } while (0)

// Tuning aids: the ONLY way this library reads the environment.  vn_knob(name, default) returns the integer value of the
// environment variable `name` if it is set (and listed in abi.hip's table: an unlisted name always returns the default),
// else `default`; callers keep the result in a function-local static (read once per process).  vn_build_info() reports
// every listed variable that is set, so a stray VN_* in the environment shows up in bench.py's JSON line.
int vn_knob(const char *name, int dflt);
// fp32x3: a conv weight operand with rows of K channels holds hi / lo bf16 granules (vn_pack_weight, VN_F32X3) when K is a
// whole number of 32-channel chunks; other K (the heads' 16-column data-gradient operand) stay fp32 and are split in registers.
// The packer and the convolution entry points both ask here.
inline bool vn_x3_presplit(int K) { return K > 0 && K % 32 == 0; }

static inline hipStream_t vn_stream(vnStream s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t vn_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t vn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Wave priority of the kernels on the step's dependency chain (convolutions, BatchNorm passes, VFE, loss): the weight
// gradients of the side stream keep the default 0.  Two waves of different kernels on one SIMD are arbitrated by priority,
// then AGE (MI355X_MICROARCH.md, "Two waves per SIMD"): the long-lived weight-gradient waves are always the older ones, so
// at equal priority the chain's short kernels lose every issue slot they contend for.  -DVN_MAIN_PRIO=n builds set
// s_setprio n at the top of those kernels (round 4 A/B; 0 / undefined = no instruction).
#if defined(VN_MAIN_PRIO) && VN_MAIN_PRIO > 0
#define VN_PRIO_MAIN() __builtin_amdgcn_s_setprio(VN_MAIN_PRIO)
#else
#define VN_PRIO_MAIN() ((void)0)
#endif

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// fp32 -> bf16 round-to-nearest-even via the hardware cast (keeps NaN a NaN)
__device__ __forceinline__ bf16_t vn_f2bf(float x) { return (bf16_t)x; }
__device__ __forceinline__ float vn_bf2f(bf16_t x) { return (float)x; }

// bf16x3 split: x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void vn_split_bf16(float x, bf16_t &hi, bf16_t &lo) {
    hi = (bf16_t)x;
    lo = (bf16_t)(x - (float)hi);
}

// "fp32x3" products (vnDtype VN_F32X3: fp32 storage, every product as three bf16 MFMAs): eight fp32 operands of a lane ->
// hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in fp32); a.b ~= ah.bh + al.bh + ah.bl, error ~2^-16 per product
__device__ __forceinline__ void vn_split8(const float (&v)[8], bf16x8_t &hi, bf16x8_t &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = (bf16_t)v[e];
        hi[e] = h;
        lo[e] = (bf16_t)(v[e] - (float)h);
    }
}
__device__ __forceinline__ void vn_split8(const f32x4_t &a0, const f32x4_t &a1, bf16x8_t &hi, bf16x8_t &lo) {
    const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    vn_split8(v, hi, lo);
}
__device__ __forceinline__ f32x4_t vn_mfma_x3(const bf16x8_t &ah, const bf16x8_t &al, const bf16x8_t &bh, const bf16x8_t &bl, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}

__device__ __forceinline__ float vn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float vn_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int vn_wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// Buffer descriptor from values the compiler can PROVE wave-uniform (cdna_hip_programming.md T20): without
// the readfirstlane of the pointer halves and the size, hipcc wraps every buffer_load ... lds that uses the
// descriptor in a waterfall loop (v_readfirstlane x4 + s_and_saveexec + loop), ~15 instructions and a
// serialisation point per load.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vn_uniform_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}
