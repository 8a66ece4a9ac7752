// A3 - query selection: per-anchor score (max over classes) + top-K anchors per image, in one kernel.
//
// Reference: DFINETransformer._select_topk (src/d_fine/arch/dfine_decoder.py:875-910):
// torch.topk(outputs_logits.max(-1).values, 300) = a reduce kernel over [B, 8400, C] followed by
// ATen's multi-pass radix top-k + sort.  Here one 1024-thread block owns one image: every thread
// keeps its anchors' scores in registers as order-preserving uint keys, an 8-bit-per-pass LDS
// histogram radix select finds the K-th largest key, the <= K survivors are compacted into LDS and
// bitonic-sorted (score descending, index ascending on ties - torch leaves tie order unspecified).
// Indices are bit-identical to torch.topk whenever the top-K scores are distinct.
#include "common.h"

namespace dfine {

constexpr int kTkThreads = 1024;
constexpr int kTkPerThread = 16;          // up to 16384 anchors per image
constexpr int kTkSort = 1024;             // K <= 1024

__device__ __forceinline__ uint32_t f2key(float f) {      // larger float -> larger key; NaN sorts last
    if (f != f) return 0u;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename T>
__global__ __launch_bounds__(kTkThreads) void topk_anchor_kernel(const T *__restrict__ logits, int64_t sb, int64_t sq,
                                                                 int Q, int C, int K, int64_t *__restrict__ out_idx,
                                                                 float *__restrict__ out_score) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_need;
    __shared__ uint32_t s_cnt;
    __shared__ uint64_t s_items[kTkSort];      // (key << 32) | (0xffffffff - index): sorts by key desc then index asc
    const int b = blockIdx.x, tid = threadIdx.x;
    const T *base = logits + (int64_t)b * sb;
    uint32_t keys[kTkPerThread];
#pragma unroll
    for (int i = 0; i < kTkPerThread; ++i) {
        const int q = tid + i * kTkThreads;
        uint32_t k = 0u;
        if (q < Q) {
            const T *p = base + (int64_t)q * sq;
            float m;
            if (sizeof(T) == 2 && (C & 7) == 0 && (sq & 7) == 0 && (sb & 7) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
                // 16-byte loads: a lane walks its own row (rows are 2 C bytes apart), so the instruction count, not
                // coalescing, is what can be saved - 8x fewer loads than element-wise
                const uint4 *pv = reinterpret_cast<const uint4 *>(p);
                m = -INFINITY;
                for (int c = 0; c < C / 8; ++c) {
                    const uint4 v = pv[c];
                    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        m = fmaxf(m, fmaxf(__uint_as_float(u[j] << 16), __uint_as_float(u[j] & 0xffff0000u)));
                }
            } else {
                m = load_f(p);
                for (int c = 1; c < C; ++c) m = fmaxf(m, load_f(p + c));
            }
            k = f2key(m);
            if (k == 0u) k = 1u;                // keep 0 for "no element"
        }
        keys[i] = k;
    }
    // ---- radix select of the K-th largest key --------------------------------------------------
    uint32_t prefix = 0u, need = (uint32_t)K;    // keys matching `prefix` on the bits decided so far
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const uint32_t mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
        for (int i = 0; i < kTkPerThread; ++i)
            if (keys[i] != 0u && (keys[i] & mask) == (prefix & mask)) atomicAdd(&hist[(keys[i] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0u; int d = 255;
            for (; d > 0; --d) { if (acc + hist[d] >= need) break; acc += hist[d]; }
            s_prefix = prefix | ((uint32_t)d << shift);
            s_need = need - acc;                  // how many keys of this digit are still needed
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        __syncthreads();
    }
    const uint32_t kth = prefix;                  // the K-th largest key; `need` of the keys equal to it are wanted
    // ---- compact: all keys > kth, plus the `need` lowest-index keys == kth ----------------------
    if (tid == 0) s_cnt = 0u;
    for (int i = tid; i < kTkSort; i += kTkThreads) s_items[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kTkPerThread; ++i) {
        const int q = tid + i * kTkThreads;
        if (keys[i] > kth) {
            const uint32_t pos = atomicAdd(&s_cnt, 1u);
            s_items[pos] = ((uint64_t)keys[i] << 32) | (uint64_t)(0xffffffffu - (uint32_t)q);
        }
    }
    __syncthreads();
    const uint32_t n_gt = s_cnt;
    // equal keys: rank by index (ascending) with a simple count of smaller indices holding the same key
    __syncthreads();
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    // gather candidates with key == kth into the tail region, then keep the `need` smallest indices
    __shared__ uint32_t s_eq[kTkSort];
#pragma unroll
    for (int i = 0; i < kTkPerThread; ++i) {
        const int q = tid + i * kTkThreads;
        if (keys[i] == kth && keys[i] != 0u) {
            const uint32_t pos = atomicAdd(&s_cnt, 1u);
            if (pos < kTkSort) s_eq[pos] = (uint32_t)q;
        }
    }
    __syncthreads();
    const uint32_t n_eq = min(s_cnt, (uint32_t)kTkSort);
    for (uint32_t i = tid; i < n_eq; i += kTkThreads) {
        const uint32_t q = s_eq[i];
        uint32_t rank = 0u;
        for (uint32_t j = 0; j < n_eq; ++j) rank += s_eq[j] < q ? 1u : 0u;
        if (rank < need && n_gt + rank < (uint32_t)kTkSort)
            s_items[n_gt + rank] = ((uint64_t)kth << 32) | (uint64_t)(0xffffffffu - q);
    }
    __syncthreads();
    // ---- bitonic sort (descending) of kTkSort 64-bit items ---------------------------------------
    for (int size = 2; size <= kTkSort; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int i = tid;
            const int j = i ^ stride;
            if (j > i) {
                const bool desc = (i & size) == 0;
                const uint64_t a = s_items[i], c = s_items[j];
                if ((a < c) == desc) { s_items[i] = c; s_items[j] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < K; i += kTkThreads) {
        const uint64_t it = s_items[i];
        const uint32_t q = 0xffffffffu - (uint32_t)(it & 0xffffffffull);
        out_idx[(int64_t)b * K + i] = (int64_t)q;
        if (out_score) {
            const uint32_t key = (uint32_t)(it >> 32);
            const uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
            out_score[(int64_t)b * K + i] = __uint_as_float(u);
        }
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

// logits [B, Q, C] dtype (element strides sb, sq; unit class stride).  out_idx [B, K] i64 (descending
// score), out_score [B, K] f32 or NULL.  Q <= 16384, K <= 1024, K <= Q.
int dfine_topk_anchors(const void *logits, int64_t sb, int64_t sq, int64_t *out_idx, float *out_score, int dtype,
                       int B, int Q, int C, int K, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!logits || !out_idx || Q < 1 || C < 1 || K < 1 || K > Q || K > kTkSort || Q > kTkThreads * kTkPerThread)
        return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(topk_anchor_kernel<float>, dim3(B), dim3(kTkThreads), 0, st, (const float *)logits, sb, sq, Q, C, K,
                           out_idx, out_score);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(topk_anchor_kernel<uint16_t>, dim3(B), dim3(kTkThreads), 0, st, (const uint16_t *)logits, sb, sq, Q,
                           C, K, out_idx, out_score);
    else return DFINE_E_BADARG;
    return check_launch();
}

}  // extern "C"
