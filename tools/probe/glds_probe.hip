// Probe: semantics of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950:
// LDS destination = (wave-uniform) pointer + lane * 16; the global source address is per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint4 *src, uint4 *out) {
    __shared__ __attribute__((aligned(16))) uint4 tile[4 * 64];      // 4 waves x 1 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int perm = (lane * 7 + 3) & 63;                              // per-lane SOURCE permutation
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + wave * 64 + perm),
                                     (__attribute__((address_space(3))) void *)(tile + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = tile[threadIdx.x];
}

int main() {
    std::vector<uint4> h(256);
    for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
    uint4 *d, *o;
    hipMalloc(&d, 256 * 16); hipMalloc(&o, 256 * 16);
    hipMemcpy(d, h.data(), 256 * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, o);
    std::vector<uint4> r(256);
    hipMemcpy(r.data(), o, 256 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63, wave = t >> 6, perm = (lane * 7 + 3) & 63;
        if (r[t].x != (unsigned)(wave * 64 + perm) || r[t].w != (unsigned)(wave * 64 + perm) * 4) ++bad;
    }
    printf("glds probe: %d mismatches of 256 (0 = LDS[base + lane*16] <- src[per-lane address])\n", bad);
    return bad != 0;
}
