// Probe: throughput of LDS float accumulation on gfx950 - ds_add_f32 (atomic, no return) against a plain
// ds_read_b32 / v_add / ds_write_b32 sequence, 64 lanes on 64 consecutive dwords, addresses changing per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip && ./lds_atomic_probe
// Measured on MI355X (round 2): ds_add_f32 203 G lane-ops/s chip-wide = 0.33 lanes/clk/CU (~770 clk per wave
// instruction with 4 waves per CU), the same rate as the L2 float atomics of the deformable-attention backward
// (196 G/s); read+add+write 1.7-4.8 T lane-ops/s (2.8-7.9 lanes/clk/CU, ~80-90 clk per dependent iteration).
// Consequence: moving a float scatter-add from L2 atomics to LDS atomics buys nothing on gfx950; an LDS accumulation
// must own its addresses (one wave per region, plain read-modify-write).  Three atomic-free rewrites of the
// deformable-attention backward built on that (tile scan + quadrant-owned RMW: 547 us; precomputed hit entries:
// 622 us; counting sort by 8x8 block + wave-per-block: 757 us, dominated by 2.2 M integer L2 atomics of the sort
// itself at ~18 G line-ops/s) did not beat the 690 us of the f32-atomic kernel by enough to replace it.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, long long *clk) {
    __shared__ float tile[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tile[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned idx = wave * 977u;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;
        const int cell = (idx >> 8) & 127;                 // wave-uniform pseudo-random 64-float row
        float *p = tile + cell * 64 + lane;
        if (MODE == 0) {
            atomicAdd(p, 1.0f);
        } else if (MODE == 1) {
            *p += 1.0f;                                      // read, add, write (one wave per row range would be needed for safety)
        } else {
            // 2 half-waves on separate rows (like two hits of 32 channels)
            float *q = tile + ((cell + (lane >> 5) * 37) & 127) * 64 + (lane & 31);
            atomicAdd(q, 1.0f);
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 256) s += tile[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

int main() {
    float *o; long long *c;
    hipMalloc(&o, 4096 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 4096;
    for (int mode = 0; mode < 3; ++mode) {
        for (int blocks : {256, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, o, iters, c);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, o, iters, c);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, o, iters, c);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long clk; hipMemcpy(&clk, c, 8, hipMemcpyDeviceToHost);
            const double lane_ops = (double)blocks * 256 * iters;
            printf("mode %d (%s) blocks %4d: %.3f ms, %.1f G lane-ops/s, %.2f lane-ops/clk/CU (2.4 GHz, 256 CUs), wave0 %lld clk for %d iters = %.1f clk/iter\n",
                   mode, mode == 0 ? "ds_add_f32" : mode == 1 ? "read+add+write" : "ds_add_f32 2 rows", blocks, ms,
                   lane_ops / ms / 1e6, lane_ops / (ms * 1e-3 * 2.4e9 * 256), clk, iters, (double)clk / iters);
        }
    }
    return 0;
}
