// Probe: how many bytes per clock a CU can pull through its vector-memory path, by source (L2-resident vs HBM) and by
// destination (LDS-DMA `global_load_lds_dwordx4` ring as in conv1x1_glds_kernel vs plain `global_load_dwordx4` into VGPRs),
// with no compute at all.  Answers whether the ~10 B/clk/CU the 1x1 kernels sit at is a property of the path or of the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o loadpath_probe loadpath_probe.hip && ./loadpath_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// MODE 0: LDS-DMA ring of 3 slots x STAGE bytes, PIECES 1-KiB pieces per wave and stage, counted vmcnt + raw barrier.
template <int WAVES, int PIECES>
__global__ __launch_bounds__(WAVES * 64) void glds_ring(const unsigned char *src, size_t region, size_t wg_stride, int stages, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int STAGE = WAVES * PIECES * 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned char *base = src + (size_t)blockIdx.x * wg_stride;
    auto issue = [&](int s) {
        const size_t off = ((size_t)s * STAGE) % region;
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            glds16(base + off + (size_t)(wave * PIECES + j) * 1024 + lane * 16,
                   __builtin_amdgcn_readfirstlane(lds0 + (s % 3) * STAGE + (wave * PIECES + j) * 1024));
    };
    issue(0);
    issue(1);
    for (int s = 0; s < stages; ++s) {
        if (s + 1 < stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < stages) issue(s + 2);
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[(blockIdx.x * 16) % STAGE];
}

// MODE 1: plain 16-byte loads into registers, UNROLL loads in flight per lane.
template <int WAVES, int UNROLL>
__global__ __launch_bounds__(WAVES * 64) void vgpr_stream(const unsigned char *src, size_t region, size_t wg_stride, int iters, unsigned *sink) {
    const unsigned char *base = src + (size_t)blockIdx.x * wg_stride;
    uint4 acc = make_uint4(0, 0, 0, 0);
    constexpr size_t STEP = (size_t)WAVES * 64 * 16;
    for (int it = 0; it < iters; ++it) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t off = (((size_t)it * UNROLL + u) * STEP) % region;
            v[u] = *reinterpret_cast<const uint4 *>(base + off + threadIdx.x * 16);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x] = acc.x;
}

template <typename F> static float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main() {
    const size_t total = ((size_t)2 << 30) + ((size_t)64 << 20);   // 2 GiB source + slack for the last stage past a region
    unsigned char *src; unsigned *sink;
    hipMalloc(&src, total); hipMalloc(&sink, 1 << 20);
    hipMemset(src, 1, total);
    int cus = 256;
    printf("%-58s %10s %12s %14s\n", "case", "ms", "TB/s", "B/clk/CU@2.4G");
    auto report = [&](const char *name, float ms, double bytes) {
        printf("%-58s %10.3f %12.2f %14.1f\n", name, ms, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / cus / 2.4e9);
    };
    // ---- LDS-DMA ring -------------------------------------------------------------------------------------------------
#define RING(WAVES, PIECES, WGS, REGION, STRIDE, NAME)                                                                        \
    {                                                                                                                         \
        constexpr int STAGE = WAVES * PIECES * 1024;                                                                          \
        const int stages = 256;                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void *>(glds_ring<WAVES, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE); \
        float ms = time_ms([&] { hipLaunchKernelGGL((glds_ring<WAVES, PIECES>), dim3(WGS), dim3(WAVES * 64), 3 * STAGE, 0, src, (size_t)(REGION), (size_t)(STRIDE), stages, sink); }); \
        report(NAME, ms, (double)(WGS) * stages * STAGE);                                                                     \
    }
    RING(8, 6, 256, 1 << 20, 0, "glds ring 8 waves x 6 KiB/stage(48K), 1 WG/CU, L2 (1 MiB shared)")
    RING(8, 6, 256, 4 << 20, 4 << 20, "glds ring 8 waves 48K stages, 1 WG/CU, HBM (4 MiB per WG)")
    RING(8, 3, 512, 1 << 20, 0, "glds ring 8 waves 24K stages, 2 WG/CU, L2")
    RING(8, 3, 512, 4 << 20, 4 << 20, "glds ring 8 waves 24K stages, 2 WG/CU, HBM")
    RING(4, 4, 1024, 1 << 20, 0, "glds ring 4 waves 16K stages, 3-4 WG/CU, L2")
    RING(4, 4, 1024, 2 << 20, 2 << 20, "glds ring 4 waves 16K stages, 3-4 WG/CU, HBM")
    RING(8, 2, 768, 1 << 20, 0, "glds ring 8 waves 16K stages, 3 WG/CU, L2")
    RING(8, 2, 768, 2 << 20, 2 << 20, "glds ring 8 waves 16K stages, 3 WG/CU, HBM")
    // ---- VGPR streams -------------------------------------------------------------------------------------------------
#define VG(WAVES, UNROLL, WGS, REGION, STRIDE, NAME)                                                                          \
    {                                                                                                                         \
        const int iters = 512 / UNROLL;                                                                                       \
        float ms = time_ms([&] { hipLaunchKernelGGL((vgpr_stream<WAVES, UNROLL>), dim3(WGS), dim3(WAVES * 64), 0, 0, src, (size_t)(REGION), (size_t)(STRIDE), iters, sink); }); \
        report(NAME, ms, (double)(WGS) * iters * UNROLL * WAVES * 64 * 16);                                                   \
    }
    VG(4, 8, 2048, 1 << 20, 0, "vgpr x4 8 in flight, 256 thr, 8 WG/CU, L2 (1 MiB shared)")
    VG(4, 8, 2048, 1 << 20, 1 << 20, "vgpr x4 8 in flight, 256 thr, 8 WG/CU, HBM")
    VG(4, 16, 1024, 1 << 20, 0, "vgpr x4 16 in flight, 256 thr, 4 WG/CU, L2")
    VG(8, 4, 256, 1 << 20, 0, "vgpr x4 4 in flight, 512 thr, 1 WG/CU, L2")
    VG(8, 8, 256, 1 << 20, 0, "vgpr x4 8 in flight, 512 thr, 1 WG/CU, L2")
    VG(8, 16, 256, 1 << 20, 0, "vgpr x4 16 in flight, 512 thr, 1 WG/CU, L2")
    VG(8, 16, 256, 4 << 20, 4 << 20, "vgpr x4 16 in flight, 512 thr, 1 WG/CU, HBM")
    return 0;
}
