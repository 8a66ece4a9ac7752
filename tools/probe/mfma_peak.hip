// Sustained matrix-core rate of this chip with nothing but MFMAs in flight (no memory traffic): the ceiling any GEMM here can
// reach, per instruction shape.   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o tools/probe/mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
    f32x4 d4 = {0}, d5 = {0}, d6 = {0}, d7 = {0};
    for (int i = 0; i < iters; ++i) {                // inline assembly: the compiler's own scheduling of MFMA chains adds moves / nops
        if (MODE == 0) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n v_mfma_f32_32x32x2_f32 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        } else if (MODE == 2) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_bf16 %1, %8, %9, %1\n"
                         "v_mfma_f32_16x16x32_bf16 %2, %8, %9, %2\n v_mfma_f32_16x16x32_bf16 %3, %8, %9, %3\n"
                         "v_mfma_f32_16x16x32_bf16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_bf16 %5, %8, %9, %5\n"
                         "v_mfma_f32_16x16x32_bf16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_bf16 %7, %8, %9, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(ha), "v"(hb));
        } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(ha), "v"(hb));
        }
    }
    for (int i = 0; i < 4; ++i) d0[i] += d4[i] + d5[i] + d6[i] + d7[i];
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
    if (s == 123.456f) out[0] = s;
}

template <int MODE> void run(const char *name, double flop_per_mfma, int waves_per_simd) {
    float *out; hipMalloc(&out, 4);
    const int iters = 20000, blocks = 256 * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * (MODE == 2 ? 8 : 4) * flop_per_mfma;
    printf("%-28s %d wave(s)/SIMD: %8.1f TFLOP/s  (%.2f ms)\n", name, waves_per_simd, fl / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_mfma_f32_32x32x2_f32", 4096, w);
        run<1>("v_mfma_f32_16x16x4_f32", 2048, w);
        run<2>("v_mfma_f32_16x16x32_bf16", 16384, w);
        run<3>("v_mfma_f32_32x32x16_bf16", 32768, w);
    }
    return 0;
}
