// A8 - fine-grained distribution refinement head: Integral + distance2bbox + LQE statistics in one
// kernel (forward) and one kernel (backward).
//
// Reference: Integral.forward (softmax over reg_max+1 bins . W(n)), distance2bbox and the first half of
// LQE.forward (softmax -> top-k probabilities + their mean) - src/d_fine/arch/dfine_decoder.py:
// 291-295,307-311 and src/d_fine/arch/utils.py:119-142 - about 25 small ATen kernels per decoder
// layer and direction.  Here one thread owns one (query, edge) row of NB = reg_max + 1 logits:
//   p = softmax(x);  d = sum_j p_j W_j;  top-K of p (descending) and their mean  -> stat
//   box (cxcywh) from the 4 edge distances around the reference box (the 4 edge lanes of a query
//   are adjacent lanes: wave shuffles, no LDS)
// backward: dx_j = p_j (g_j - sum_i g_i p_i) with g_j = gd * W_j + [j in top-K] (g_top[rank] + g_mean / K).
#include "common.h"

namespace dfine {

struct FdrTable { float w[64]; float reg_scale; };

template <typename T, int NB, int K>
__global__ __launch_bounds__(256) void fdr_fwd_kernel(const T *__restrict__ corners, const float *__restrict__ ref,
                                                       FdrTable tab, float *__restrict__ boxes,
                                                       float *__restrict__ stat, uint8_t *__restrict__ top_idx,
                                                       int N) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool live = r < N * 4;
    const int n = live ? r >> 2 : 0, e = r & 3;
    const T *xp = corners + (int64_t)n * 4 * NB + e * NB;
    float p[NB], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NB; ++j) { p[j] = load_f(xp + j); mx = fmaxf(mx, p[j]); }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) { p[j] = __expf(p[j] - mx); s += p[j]; }
    const float inv = 1.f / s;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) { p[j] *= inv; d += p[j] * tab.w[j]; }
    // top-K (descending, first index wins ties)
    float tsum = 0.f;
    unsigned taken_lo = 0u, taken_hi = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float best = -1.f; int bi = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool used = j < 32 ? (taken_lo >> j) & 1u : (taken_hi >> (j - 32)) & 1u;
            if (!used && p[j] > best) { best = p[j]; bi = j; }
        }
        if (bi < 32) taken_lo |= 1u << bi; else taken_hi |= 1u << (bi - 32);
        tsum += best;
        if (live) {
            stat[(int64_t)n * 4 * (K + 1) + e * (K + 1) + k] = best;
            top_idx[(int64_t)r * K + k] = (uint8_t)bi;
        }
    }
    if (live) stat[(int64_t)n * 4 * (K + 1) + e * (K + 1) + K] = tsum / (float)K;
    // distance2bbox: gather the 4 edge distances of the query on its first lane
    const int base = (threadIdx.x & 63) & ~3;
    const float d0 = __shfl(d, base, 64), d1 = __shfl(d, base + 1, 64), d2 = __shfl(d, base + 2, 64),
                d3 = __shfl(d, base + 3, 64);
    if (live && e == 0) {
        const float4 rb = *reinterpret_cast<const float4 *>(ref + (int64_t)n * 4);
        const float rs = fabsf(tab.reg_scale);
        const float sx = rb.z / rs, sy = rb.w / rs;
        const float x1 = rb.x - (0.5f * rs + d0) * sx, y1 = rb.y - (0.5f * rs + d1) * sy;
        const float x2 = rb.x + (0.5f * rs + d2) * sx, y2 = rb.y + (0.5f * rs + d3) * sy;
        *reinterpret_cast<float4 *>(boxes + (int64_t)n * 4) = make_float4((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1);
    }
}

template <typename T, int NB, int K>
__global__ __launch_bounds__(256) void fdr_bwd_kernel(const T *__restrict__ corners, const float *__restrict__ ref,
                                                       FdrTable tab, const float *__restrict__ g_boxes,
                                                       const float *__restrict__ g_stat,
                                                       const uint8_t *__restrict__ top_idx, T *__restrict__ g_corners,
                                                       int N) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= N * 4) return;
    const int n = r >> 2, e = r & 3;
    const T *xp = corners + (int64_t)n * 4 * NB + e * NB;
    float p[NB], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NB; ++j) { p[j] = load_f(xp + j); mx = fmaxf(mx, p[j]); }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) { p[j] = __expf(p[j] - mx); s += p[j]; }
    const float inv = 1.f / s;
    // gradient wrt the edge distance from the box gradient
    float gd = 0.f;
    if (g_boxes) {
        const float4 gb = *reinterpret_cast<const float4 *>(g_boxes + (int64_t)n * 4);
        const float4 rb = *reinterpret_cast<const float4 *>(ref + (int64_t)n * 4);
        const float rs = fabsf(tab.reg_scale);
        const float sx = rb.z / rs, sy = rb.w / rs;
        gd = e == 0 ? (-0.5f * gb.x + gb.z) * sx : e == 1 ? (-0.5f * gb.y + gb.w) * sy
           : e == 2 ? (0.5f * gb.x + gb.z) * sx : (0.5f * gb.y + gb.w) * sy;
    }
    float g[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { p[j] *= inv; g[j] = gd * tab.w[j]; }
    if (g_stat) {
        const float *gs = g_stat + (int64_t)n * 4 * (K + 1) + e * (K + 1);
        const float gm = gs[K] / (float)K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int bi = top_idx[(int64_t)r * K + k];
            const float add = gs[k] + gm;
#pragma unroll
            for (int j = 0; j < NB; ++j) g[j] += j == bi ? add : 0.f;
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) dot += g[j] * p[j];
    T *gp = g_corners + (int64_t)n * 4 * NB + e * NB;
#pragma unroll
    for (int j = 0; j < NB; ++j) store_f(gp + j, p[j] * (g[j] - dot));
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_fdr_fwd(const void *corners, const float *ref, const float *wtable, float reg_scale, float *boxes,
                  float *stat, uint8_t *top_idx, int dtype, int N, int reg_max, int K, void *stream) {
    if (N == 0) return DFINE_OK;
    if (!corners || !ref || !wtable || !boxes || !stat || !top_idx || reg_max != 32 || K != 4) return DFINE_E_BADARG;
    FdrTable tab;
    for (int j = 0; j <= reg_max; ++j) tab.w[j] = wtable[j];
    tab.reg_scale = reg_scale;
    const int blocks = (N * 4 + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL((fdr_fwd_kernel<float, 33, 4>), dim3(blocks), dim3(256), 0, st, (const float *)corners, ref, tab, boxes,
                           stat, top_idx, N);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL((fdr_fwd_kernel<uint16_t, 33, 4>), dim3(blocks), dim3(256), 0, st, (const uint16_t *)corners, ref, tab,
                           boxes, stat, top_idx, N);
    else return DFINE_E_BADARG;
    return check_launch();
}

int dfine_fdr_bwd(const void *corners, const float *ref, const float *wtable, float reg_scale, const float *g_boxes,
                  const float *g_stat, const uint8_t *top_idx, void *g_corners, int dtype, int N, int reg_max, int K,
                  void *stream) {
    if (N == 0) return DFINE_OK;
    if (!corners || !ref || !wtable || !top_idx || !g_corners || reg_max != 32 || K != 4) return DFINE_E_BADARG;
    FdrTable tab;
    for (int j = 0; j <= reg_max; ++j) tab.w[j] = wtable[j];
    tab.reg_scale = reg_scale;
    const int blocks = (N * 4 + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL((fdr_bwd_kernel<float, 33, 4>), dim3(blocks), dim3(256), 0, st, (const float *)corners, ref, tab, g_boxes,
                           g_stat, top_idx, (float *)g_corners, N);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL((fdr_bwd_kernel<uint16_t, 33, 4>), dim3(blocks), dim3(256), 0, st, (const uint16_t *)corners, ref, tab,
                           g_boxes, g_stat, top_idx, (uint16_t *)g_corners, N);
    else return DFINE_E_BADARG;
    return check_launch();
}

}  // extern "C"
