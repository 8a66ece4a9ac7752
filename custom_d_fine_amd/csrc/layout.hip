// A3 - encoder memory layout: the decoder works on tokens [B, L, C] (L = all pixels of all levels), the encoder on maps
// [B, C, H, W].  Reference: DFINETransformer._get_encoder_input flattens, permutes and concatenates the levels
// (src/d_fine/arch/dfine_decoder.py:778-801) - a strided read inside torch.cat forward, and in the backward strided
// gradient views that every consumer has to add to / copy from element by element.  Here both directions are tiled
// transposes through LDS with 16-byte accesses on both sides: maps -> tokens writes each level at its row offset of the
// memory tensor, tokens -> maps returns one CONTIGUOUS map per level.
#include "common.h"

namespace dfine {

constexpr int kTT = 64;                  // tile: 64 channels x 64 pixels
constexpr int kTThreads = 256;

// src [B][C][HW] (plane-contiguous) -> dst [B][L][C] rows [row0, row0 + HW)
__global__ __launch_bounds__(kTThreads) void maps_to_tokens_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                                  int C, int HW, int L, int row0) {
    __shared__ uint16_t tile[kTT][kTT + 8];                // [channel][pixel], pitch 72: 16-byte rows, shifted banks
    const int b = blockIdx.z, c0 = blockIdx.y * kTT, p0 = blockIdx.x * kTT;
    const uint16_t *sb = src + ((int64_t)b * C + c0) * HW + p0;
    // load: 64 channels x 64 pixels, 8 pixels (16 B) per thread, 2 passes
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 32 + threadIdx.x / 8, v = (threadIdx.x % 8) * 8;
        uint4 d = make_uint4(0, 0, 0, 0);
        if (c0 + r < C && p0 + v < HW) d = *reinterpret_cast<const uint4 *>(sb + (int64_t)r * HW + v);
        *reinterpret_cast<uint4 *>(&tile[r][v]) = d;
    }
    __syncthreads();
    uint16_t *db = dst + ((int64_t)b * L + row0 + p0) * C + c0;
    // store: 64 pixels x 64 channels, 8 channels (16 B) per thread
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 32 + threadIdx.x / 8, cv = (threadIdx.x % 8) * 8;
        if (p0 + p < HW && c0 + cv < C) {
            uint16_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = tile[cv + k][p];
            uint4 o;
            o.x = e[0] | ((uint32_t)e[1] << 16); o.y = e[2] | ((uint32_t)e[3] << 16);
            o.z = e[4] | ((uint32_t)e[5] << 16); o.w = e[6] | ((uint32_t)e[7] << 16);
            *reinterpret_cast<uint4 *>(db + (int64_t)p * C + cv) = o;
        }
    }
}

// src [B][L][C] rows [row0, row0 + HW) -> dst [B][C][HW]
__global__ __launch_bounds__(kTThreads) void tokens_to_maps_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                                  int C, int HW, int L, int row0) {
    __shared__ uint16_t tile[kTT][kTT + 8];                // [pixel][channel]
    const int b = blockIdx.z, c0 = blockIdx.y * kTT, p0 = blockIdx.x * kTT;
    const uint16_t *sb = src + ((int64_t)b * L + row0 + p0) * C + c0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 32 + threadIdx.x / 8, cv = (threadIdx.x % 8) * 8;
        uint4 d = make_uint4(0, 0, 0, 0);
        if (p0 + p < HW && c0 + cv < C) d = *reinterpret_cast<const uint4 *>(sb + (int64_t)p * C + cv);
        *reinterpret_cast<uint4 *>(&tile[p][cv]) = d;
    }
    __syncthreads();
    uint16_t *db = dst + ((int64_t)b * C + c0) * HW + p0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 32 + threadIdx.x / 8, v = (threadIdx.x % 8) * 8;
        if (c0 + r < C && p0 + v < HW) {
            uint16_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = tile[v + k][r];
            uint4 o;
            o.x = e[0] | ((uint32_t)e[1] << 16); o.y = e[2] | ((uint32_t)e[3] << 16);
            o.z = e[4] | ((uint32_t)e[5] << 16); o.w = e[6] | ((uint32_t)e[7] << 16);
            *reinterpret_cast<uint4 *>(db + (int64_t)r * HW + v) = o;
        }
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

// map [B, C, HW] bf16 -> rows [row0, row0 + HW) of tokens [B, L, C]; to_tokens = 0: the inverse.  C % 8 == 0, HW % 8 == 0.
int dfine_maps_tokens_bf16(const void *map, void *tokens, int B, int C, int HW, int L, int row0, int to_tokens, void *stream) {
    if (B == 0 || C == 0 || HW == 0) return DFINE_OK;
    if (!map || !tokens || B < 0 || C < 8 || (C & 7) || HW < 8 || (HW & 7) || row0 < 0 || row0 + HW > L) return DFINE_E_BADARG;
    dim3 grid((HW + kTT - 1) / kTT, (C + kTT - 1) / kTT, B);
    if (to_tokens)
        hipLaunchKernelGGL(maps_to_tokens_kernel, grid, dim3(kTThreads), 0, (hipStream_t)stream, (const uint16_t *)map,
                           (uint16_t *)tokens, C, HW, L, row0);
    else
        hipLaunchKernelGGL(tokens_to_maps_kernel, grid, dim3(kTThreads), 0, (hipStream_t)stream, (const uint16_t *)tokens,
                           (uint16_t *)const_cast<void *>(map), C, HW, L, row0);
    return check_launch();
}

}  // extern "C"
