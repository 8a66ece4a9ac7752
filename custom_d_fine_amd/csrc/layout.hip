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

// ---- nearest-neighbour 2x upsampling of the FPN top-down path (ref hybrid_encoder.py:472 F.interpolate(scale_factor=2,
// mode="nearest")) and its backward (sum of the 2 x 2 block): pure data movement, one thread per 8 (or 4) input pixels - one
// vector load and four vector stores (two output rows x 16 pixels) forward, four loads and one store backward.
template <int V> struct UpVec;
template <> struct UpVec<8> { typedef uint4 type; };
template <> struct UpVec<4> { typedef uint2 type; };

template <int V>   // V input pixels per thread: 8 (16-byte vectors) or 4 (W % 8 != 0: the 20-wide level)
__global__ __launch_bounds__(256) void upsample2_nearest_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y,
                                                                int64_t nvec, int WV /* W / V */, int H) {
    typedef typename UpVec<V>::type vec_t;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t row = v / WV;                      // plane * H + h
        const int wv = (int)(v - row * WV);
        const vec_t a = *reinterpret_cast<const vec_t *>(x + v * V);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&a);
        uint32_t d[V];
#pragma unroll
        for (int k = 0; k < V / 2; ++k) {
            d[2 * k] = (w[k] & 0xffffu) | (w[k] << 16);             // pixel 2k twice
            d[2 * k + 1] = (w[k] >> 16) | (w[k] & 0xffff0000u);     // pixel 2k + 1 twice
        }
        const int64_t plane = row / H;
        const int h = (int)(row - plane * H);
        const int64_t pitch = (int64_t)WV * 2 * V;                  // output row length
        uint16_t *o = y + (plane * 2 * H + 2 * h) * pitch + (int64_t)wv * 2 * V;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                vec_t ov;
                uint32_t *op = reinterpret_cast<uint32_t *>(&ov);
#pragma unroll
                for (int k = 0; k < V / 2; ++k) op[k] = d[half * (V / 2) + k];
                *reinterpret_cast<vec_t *>(o + r * pitch + half * V) = ov;
            }
    }
}

template <int V>
__global__ __launch_bounds__(256) void upsample2_nearest_bwd_kernel(const uint16_t *__restrict__ dy, uint16_t *__restrict__ dx,
                                                                    int64_t nvec, int WV, int H) {
    typedef typename UpVec<V>::type vec_t;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t row = v / WV;
        const int wv = (int)(v - row * WV);
        const int64_t plane = row / H;
        const int h = (int)(row - plane * H);
        const int64_t pitch = (int64_t)WV * 2 * V;
        const uint16_t *g = dy + (plane * 2 * H + 2 * h) * pitch + (int64_t)wv * 2 * V;
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const vec_t t = *reinterpret_cast<const vec_t *>(g + r * pitch + half * V);
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
                for (int k = 0; k < V / 2; ++k)      // dword k of this half = output pixels of input pixel half * V / 2 + k
                    s[half * (V / 2) + k] += __uint_as_float(tp[k] << 16) + __uint_as_float(tp[k] & 0xffff0000u);
            }
        vec_t o;
        uint32_t *op = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
        for (int k = 0; k < V / 2; ++k) op[k] = pack_bf16x2(s[2 * k], s[2 * k + 1]);
        *reinterpret_cast<vec_t *>(dx + v * V) = o;
    }
}

// Weight gradient of an embedding lookup with FEW rows and MANY lookups (the denoising class embedding: 81 x 256 weights, B * dn =
// 6 272 lookups per step): dw[c][:] = sum over the lookups i with idx[i] == c of g[i][:].  One workgroup per (weight row, 256-column
// block) walks the index list (25 KB, L2) and adds the matching rows in order - deterministic, no sort, no atomics.  ATen's
// embedding_dense_backward sorts the indices first (rocPRIM radix sort + segmented sum_and_scatter: 170-370 us for this problem).
template <typename IT>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float *__restrict__ g, const IT *__restrict__ idx,
                                                            float *__restrict__ dw, int64_t n, int D, int padding_idx) {
    __shared__ int hits[256];
    __shared__ int nhit;
    const int c = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x;
    float acc = 0.f;
    for (int64_t i0 = 0; i0 < n; i0 += 256) {
        // the 256 lookups of this round that hit row c, compacted in order
        if (threadIdx.x == 0) nhit = 0;
        __syncthreads();
        const int64_t i = i0 + threadIdx.x;
        const bool hit = i < n && (int64_t)idx[i] == c && c != padding_idx;
        const unsigned long long m = __ballot(hit);
        __shared__ int wbase[4];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wbase[wave] = __popcll(m);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wbase[w];
        if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)(i - i0);
        if (threadIdx.x == 0) nhit = wbase[0] + wbase[1] + wbase[2] + wbase[3];
        __syncthreads();
        const int nh = nhit;
        if (col < D)
            for (int h = 0; h < nh; ++h) acc += g[(i0 + hits[h]) * D + col];
        __syncthreads();
    }
    if (col < D) dw[(int64_t)c * D + col] = acc;
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

// y [planes, 2H, 2W] = nearest-neighbour 2x upsampling of x [planes, H, W] (bf16, W % 4 == 0); backward = 0: forward,
// 1: x := sum over the 2 x 2 blocks of y (the gradient).
int dfine_upsample2_nearest_bf16(void *x, void *y, int64_t planes, int H, int W, int backward, void *stream) {
    if (planes == 0 || H == 0 || W == 0) return DFINE_OK;
    if (!x || !y || planes < 0 || H < 1 || W < 4 || (W & 3)) return DFINE_E_BADARG;
    const int V = (W & 7) ? 4 : 8;
    const int64_t nvec = planes * H * (W / V);
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (backward) {
        if (V == 8) hipLaunchKernelGGL(upsample2_nearest_bwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint16_t *)y, (uint16_t *)x, nvec, W / 8, H);
        else hipLaunchKernelGGL(upsample2_nearest_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint16_t *)y, (uint16_t *)x, nvec, W / 4, H);
    } else {
        if (V == 8) hipLaunchKernelGGL(upsample2_nearest_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint16_t *)x, (uint16_t *)y, nvec, W / 8, H);
        else hipLaunchKernelGGL(upsample2_nearest_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint16_t *)x, (uint16_t *)y, nvec, W / 4, H);
    }
    return check_launch();
}

// dw [rows, D] f32 (overwritten) = gradient of F.embedding(idx, weight [rows, D]) given g [n, D] f32 and idx [n] int32 / int64 (idx_bits; lookups of
// padding_idx, or -1 for none, contribute nothing).  Meant for small tables (rows <= a few hundred).
int dfine_embedding_bwd(const float *g, const void *idx, int idx_bits, float *dw, int64_t n, int rows, int D, int padding_idx,
                        void *stream) {
    if (rows == 0 || D == 0) return DFINE_OK;
    if (!dw || rows < 0 || D < 0 || n < 0 || (n > 0 && (!g || !idx)) || (idx_bits != 32 && idx_bits != 64)) return DFINE_E_BADARG;
    const dim3 grid(rows, (D + 255) / 256);
    if (idx_bits == 64)
        hipLaunchKernelGGL(embedding_bwd_kernel<int64_t>, grid, dim3(256), 0, (hipStream_t)stream, g, (const int64_t *)idx, dw, n, D, padding_idx);
    else
        hipLaunchKernelGGL(embedding_bwd_kernel<int32_t>, grid, dim3(256), 0, (hipStream_t)stream, g, (const int32_t *)idx, dw, n, D, padding_idx);
    return check_launch();
}

}  // extern "C"
