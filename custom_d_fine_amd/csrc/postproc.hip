// A18 - detection post-processor: sigmoid -> top-K over all Q*C (query, class) scores -> label / query split ->
// absolute xyxy boxes, one kernel, one workgroup per image.
//
// Reference: DFINEPostProcessor.forward (src/dl/export.py:61-100) and the same block inside
// Trainer.preds_postprocess (src/dl/train.py:262-277) / Torch_model._preds_postprocess
// (src/infer/torch_model.py:197-214): torch.sigmoid over [B,Q,C], torch.topk(flat, K), idx % C, idx // C, a gather of
// the converted boxes - five ATen kernels plus topk's multi-pass radix sort.  Integer outputs (labels, query index) must
// be bit-identical; the box arithmetic (export.py:35-59) is reproduced operation by operation in fp32 with contraction
// off (build.py compiles this file with -ffp-contract=off: `cx - w/2` must not become an fma).
//
// Selection runs on the LOGITS (order-preserving uint keys in registers, 8-bit radix select through an LDS histogram,
// bitonic sort of the <= 1024 survivors): sigmoid is monotone, so the order equals the reference's order on the
// sigmoid scores wherever those are distinct; among scores that tie after rounding the larger logit / lower flat index
// goes first (torch.topk leaves tie order unspecified).
#include "common.h"

namespace dfine {

constexpr int kPpThreads = 1024;
constexpr int kPpPerThread = 32;          // keys of up to 32768 (query, class) scores per image stay in registers; larger
                                          // problems (Objects365: 300 x 365) recompute them from L2 in every pass
constexpr int kPpSortSmall = 1024;        // K <= 1024: one element per thread in the bitonic network
constexpr int kPpSortLarge = 4096;        // K <= 4096

__device__ __forceinline__ uint32_t pp_key(float f) {      // larger float -> larger key; NaN sorts FIRST like torch.topk
    if (f != f) return 0xffffffffu;
    const uint32_t u = __float_as_uint(f);
    const uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return k == 0u ? 1u : k;                                // 0 is reserved for "no element"
}

__device__ __forceinline__ float pp_unkey(uint32_t key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

template <typename T, bool REG, int kPpSort>
__global__ __launch_bounds__(kPpThreads) void postprocess_kernel(const T *__restrict__ logits, const float *__restrict__ boxes,
                                                                 int Q, int C, int K, float height, float width, int to_round,
                                                                 int64_t *__restrict__ out_label, int64_t *__restrict__ out_query,
                                                                 float *__restrict__ out_box, float *__restrict__ out_score) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_need, s_cnt;
    __shared__ uint64_t s_items[kPpSort];      // (key << 32) | (0xffffffff - flat index): key desc, index asc
    __shared__ uint32_t s_eq[kPpSort];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = Q * C;
    const T *base = logits + (int64_t)b * N;
    uint32_t keys[REG ? kPpPerThread : 1];
    if (REG) {
#pragma unroll
        for (int i = 0; i < kPpPerThread; ++i) {
            const int q = tid + i * kPpThreads;
            keys[i] = q < N ? pp_key(load_f(base + q)) : 0u;
        }
    }
    // visit every (flat index, key) this thread owns: from registers, or recomputed from the (L2-resident) logits
    auto for_each_key = [&](auto &&fn) {
        if (REG) {
#pragma unroll
            for (int i = 0; i < kPpPerThread; ++i) fn(tid + i * kPpThreads, keys[i]);
        } else {
            for (int q = tid; q < N; q += kPpThreads) fn(q, pp_key(load_f(base + q)));
        }
    };
    uint32_t prefix = 0u, need = (uint32_t)K;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const uint32_t mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        for_each_key([&](int, uint32_t key) {
            if (key != 0u && (key & mask) == (prefix & mask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        });
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0u; int d = 255;
            for (; d > 0; --d) { if (acc + hist[d] >= need) break; acc += hist[d]; }
            s_prefix = prefix | ((uint32_t)d << shift);
            s_need = need - acc;
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        __syncthreads();
    }
    const uint32_t kth = prefix;
    if (tid == 0) s_cnt = 0u;
    for (int i = tid; i < kPpSort; i += kPpThreads) s_items[i] = 0ull;
    __syncthreads();
    for_each_key([&](int q, uint32_t key) {
        if (key > kth) {
            const uint32_t pos = atomicAdd(&s_cnt, 1u);
            s_items[pos] = ((uint64_t)key << 32) | (uint64_t)(0xffffffffu - (uint32_t)q);
        }
    });
    __syncthreads();
    const uint32_t n_gt = s_cnt;
    __syncthreads();
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    for_each_key([&](int q, uint32_t key) {
        if (key == kth && key != 0u) {
            const uint32_t pos = atomicAdd(&s_cnt, 1u);
            if (pos < kPpSort) s_eq[pos] = (uint32_t)q;
        }
    });
    __syncthreads();
    // keys equal to the K-th: the `need` lowest flat indices.  When more than kPpSort elements tie at the cut (a constant
    // logit map) only the first kPpSort collected are ranked - still a valid top-K of tied scores.
    const uint32_t n_eq = min(s_cnt, (uint32_t)kPpSort);
    for (uint32_t i = tid; i < n_eq; i += kPpThreads) {
        const uint32_t q = s_eq[i];
        uint32_t rank = 0u;
        for (uint32_t j = 0; j < n_eq; ++j) rank += s_eq[j] < q ? 1u : 0u;
        if (rank < need && n_gt + rank < (uint32_t)kPpSort)
            s_items[n_gt + rank] = ((uint64_t)kth << 32) | (uint64_t)(0xffffffffu - q);
    }
    __syncthreads();
    for (int size = 2; size <= kPpSort; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < kPpSort; i += kPpThreads) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const uint64_t a = s_items[i], c = s_items[j];
                    if ((a < c) == desc) { s_items[i] = c; s_items[j] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < K; i += kPpThreads) {
        const uint64_t it = s_items[i];
        const uint32_t flat = 0xffffffffu - (uint32_t)(it & 0xffffffffull);
        const int q = (int)(flat / (uint32_t)C);
        const int64_t o = (int64_t)b * K + i;
        out_label[o] = (int64_t)(flat - (uint32_t)q * (uint32_t)C);
        out_query[o] = (int64_t)q;
        const float x = pp_unkey((uint32_t)(it >> 32));
        out_score[o] = 1.0f / (1.0f + expf(-x));
        const float *bx = boxes + ((int64_t)b * Q + q) * 4;
        // export.py:37-59, same operation order
        const float xc = bx[0] * width, yc = bx[1] * height, bw = bx[2] * width, bh = bx[3] * height;
        float x0 = xc - bw / 2.f, y0 = yc - bh / 2.f, x1 = xc + bw / 2.f, y1 = yc + bh / 2.f;
        if (to_round) {
            x0 = fmaxf(floorf(x0), 1.f); y0 = fmaxf(floorf(y0), 1.f);
            x1 = fminf(ceilf(x1), width - 1.f); y1 = fminf(ceilf(y1), height - 1.f);
        } else {
            x0 = fmaxf(x0, 0.f); y0 = fmaxf(y0, 0.f);
            x1 = fminf(x1, width); y1 = fminf(y1, height);
        }
        *reinterpret_cast<float4 *>(out_box + o * 4) = make_float4(x0, y0, x1, y1);
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_postprocess(const void *logits, const float *boxes, int64_t *labels, int64_t *query_idx, float *out_boxes,
                      float *scores, int dtype, int B, int Q, int C, int K, int height, int width, int to_round,
                      void *stream) {
    if (B == 0) return DFINE_OK;
    if (!logits || !boxes || !labels || !query_idx || !out_boxes || !scores || Q < 1 || C < 1 || K < 1 ||
        (int64_t)Q * C > 0x7fffffff || K > kPpSortLarge || K > Q * C || height < 1 || width < 1)
        return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const bool small = (int64_t)Q * C <= (int64_t)kPpThreads * kPpPerThread && K <= kPpSortSmall;
#define DFINE_PP(T, REG, SORT)                                                                                             \
    hipLaunchKernelGGL((postprocess_kernel<T, REG, SORT>), dim3(B), dim3(kPpThreads), 0, st, (const T *)logits, boxes, Q, C, K, \
                       (float)height, (float)width, to_round, labels, query_idx, out_boxes, scores)
    if (dtype == DFINE_F32) { if (small) DFINE_PP(float, true, kPpSortSmall); else DFINE_PP(float, false, kPpSortLarge); }
    else if (dtype == DFINE_BF16) { if (small) DFINE_PP(uint16_t, true, kPpSortSmall); else DFINE_PP(uint16_t, false, kPpSortLarge); }
    else return DFINE_E_BADARG;
#undef DFINE_PP
    return check_launch();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// (f1) Inference pre-processing on the device: uint8 HWC BGR image(s) -> [resize | letterbox] -> RGB CHW float / 255.
// Reference: Torch_model._preprocess / _prepare_inputs (src/infer/torch_model.py:240-298) and letterbox (:378-418):
// cv2.resize(INTER_LINEAR) [+ cv2.copyMakeBorder(114)] on the host, channel flip + transpose in numpy, H2D of the uint8
// tensor, .float().div_(255) on the device.  Here the host only uploads the raw frame; one kernel does the rest.
// The bilinear resize restates OpenCV's 8-bit path (imgproc/resize.cpp, not in /root/reference - a pip dependency): pixel
// centres aligned (src = (dst + 0.5) * scale - 0.5), coefficients rounded to 11 fixed-point bits, horizontal pass in
// int32, vertical pass (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, so that the uint8 image the
// network sees is the one cv2 would have produced.
namespace dfine {

__device__ __forceinline__ void resize_coef(int d, double scale, int ssize, int *s0, int *a0, int *a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    *s0 = s;
    *a0 = (int)lrintf((1.f - f) * 2048.f);      // saturate_cast<short>: round to nearest (even), values within [0, 2048]
    *a1 = (int)lrintf(f * 2048.f);
}

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ src, T *__restrict__ dst, int B, int Hs, int Ws,
                                                         int Ho, int Wo, int rh, int rw, int top, int left, float pad_value) {
    const int64_t total = (int64_t)B * Ho * Wo;
    const double sx = (double)Ws / rw, sy = (double)Hs / rh;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((int64_t)Wo * Ho));
        float out[3] = {pad_value, pad_value, pad_value};
        const int rx = x - left, ry = y - top;
        if (rx >= 0 && rx < rw && ry >= 0 && ry < rh) {
            const uint8_t *img = src + (int64_t)b * Hs * Ws * 3;
            if (rw == Ws && rh == Hs) {
#pragma unroll
                for (int c = 0; c < 3; ++c) out[c] = (float)img[((int64_t)ry * Ws + rx) * 3 + c];
            } else {
                int x0, ax0, ax1, y0, ay0, ay1;
                resize_coef(rx, sx, Ws, &x0, &ax0, &ax1);
                resize_coef(ry, sy, Hs, &y0, &ay0, &ay1);
                const int x1 = min(x0 + 1, Ws - 1), y1 = min(y0 + 1, Hs - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int r0 = img[((int64_t)y0 * Ws + x0) * 3 + c] * ax0 + img[((int64_t)y0 * Ws + x1) * 3 + c] * ax1;
                    const int r1 = img[((int64_t)y1 * Ws + x0) * 3 + c] * ax0 + img[((int64_t)y1 * Ws + x1) * 3 + c] * ax1;
                    out[c] = (float)((((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2);
                }
            }
        }
        const int64_t plane = (int64_t)Ho * Wo, o = (int64_t)b * 3 * plane + (int64_t)y * Wo + x;
        store_f(dst + o, out[2] / 255.0f);              // BGR -> RGB
        store_f(dst + o + plane, out[1] / 255.0f);
        store_f(dst + o + 2 * plane, out[0] / 255.0f);
    }
}

}  // namespace dfine

extern "C" {

// src uint8 [B, Hs, Ws, 3] (BGR, device) -> dst [B, 3, Ho, Wo] dtype (RGB / 255): the source is resized to (rh, rw) and placed at
// (top, left) of the output, the rest is filled with pad_value / 255 (letterbox: 114).  Plain resize: rh = Ho, rw = Wo, top = left = 0.
int dfine_preprocess_u8(const uint8_t *src, void *dst, int dtype, int B, int Hs, int Ws, int Ho, int Wo, int rh, int rw, int top,
                        int left, int pad_value, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!src || !dst || Hs < 1 || Ws < 1 || Ho < 1 || Wo < 1 || rh < 1 || rw < 1 || top < 0 || left < 0 || top + rh > Ho ||
        left + rw > Wo)
        return DFINE_E_BADARG;
    const int64_t total = (int64_t)B * Ho * Wo;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(dfine::preprocess_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float *)dst, B, Hs, Ws, Ho, Wo, rh, rw,
                           top, left, (float)pad_value);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(dfine::preprocess_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, src, (uint16_t *)dst, B, Hs, Ws, Ho, Wo,
                           rh, rw, top, left, (float)pad_value);
    else return DFINE_E_BADARG;
    return dfine::check_launch();
}

}  // extern "C"
