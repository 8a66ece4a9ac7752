// Shared helpers of the gfx950 kernels (wave = 64 lanes, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dfine_hip.h"

namespace dfine {

void set_last_error(hipError_t e);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(e);
        return DFINE_E_LAUNCH;
    }
    return DFINE_OK;
}

// Zero fill as a KERNEL launch, not hipMemsetAsync: a memset recorded into a HIP graph (the captured backward segment of
// dl/engine.py) came back with garbage in the filled range from the second replay on (stem weight gradients of 1e12 .. 1e35,
// tests/test_graph_gpu.py) - kernel nodes replay faithfully.  `bytes` must be a multiple of 4, `p` 4-byte aligned.
static __global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t *__restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
inline void zero_fill_async(void *p, size_t bytes, hipStream_t st) {
    const size_t words = bytes >> 2;
    if (!words) return;
    size_t blocks = (words + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)p, words);
}

// ---- bf16 <-> f32 (round to nearest even), raw 16-bit storage ------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// gfx950 converts with round-to-nearest-even in hardware (v_cvt_pk_bf16_f32, two floats per instruction); the bit
// arithmetic it replaces costs ~6 VALU instructions per element, which matters: the element-wise kernels (BatchNorm,
// activations, epilogues) run close to the VALU roofline, not only the HBM one.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // inline asm, not __builtin_convertvector to a __bf16 vector: the vector-typed conversion made the register allocator
    // give attn_bwd_dq_kernel 332 VGPRs (one wave per SIMD, 2x slower) where this form needs 88
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, f) & 0xffffu); }

struct f32x4 { float x, y, z, w; };

// 4 consecutive channels -> fp32, for the two storage types
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float *p) {
        float4 v = *reinterpret_cast<const float4 *>(p);
        return {v.x, v.y, v.z, v.w};
    }
    static __device__ __forceinline__ void store(float *p, f32x4 v) {
        *reinterpret_cast<float4 *>(p) = make_float4(v.x, v.y, v.z, v.w);
    }
};
template <> struct Vec4<uint16_t> {
    static __device__ __forceinline__ f32x4 load(const uint16_t *p) {
        uint2 v = *reinterpret_cast<const uint2 *>(p);
        return {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
    }
    static __device__ __forceinline__ void store(uint16_t *p, f32x4 v) {
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y);
        o.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = o;
    }
};

template <typename T> __device__ __forceinline__ float load_f(const T *p);
template <> __device__ __forceinline__ float load_f<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float load_f<uint16_t>(const uint16_t *p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void store_f(T *p, float v);
template <> __device__ __forceinline__ void store_f<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_f<uint16_t>(uint16_t *p, float v) { *p = f32_to_bf16(v); }

}  // namespace dfine
