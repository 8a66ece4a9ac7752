// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element lands in (lane, j)?
// LDS image: 64 rows x 64 cols of uint16, value = row * 256 + col.  Lane l = 16 g + i supplies the address of
// row (4 g + i / 4), columns 4 (i % 4) .. +3.  Expected (guide): lane receives column (l & 15) of the 4 x 16 block
// rows 4g .. 4g+3, i.e. out[j] = (4 g + j) * 256 + (l & 15).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    const uint16_t *p = lds + (4 * g + i / 4) * 64 + 4 * (i % 4);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3))) *)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" (r%2d,c%2d)", h[l * 4 + j] / 256, h[l * 4 + j] % 256);
            if (h[l * 4 + j] != (4 * (l >> 4) + j) * 256 + (l & 15)) ++bad;
        }
        printf("\n");
    }
    printf("mismatches vs expected mapping: %d\n", bad);
    return 0;
}
