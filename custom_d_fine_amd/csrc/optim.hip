// A16 - optimizer step on flat fp32 buffers: global-norm gradient clipping + AdamW + EMA (+ the
// zero_grad of the next step) in two launches per parameter group.
//
// Reference: Trainer.optimizer_step (src/dl/train.py:512-535) = clip_grad_norm_(max_norm) ->
// AdamW.step -> zero_grad -> ModelEMA.update (src/dl/train.py:52-73; two ops per state-dict tensor,
// 2106 launches for D-FINE-m) with the four parameter groups of build_optimizer
// (src/d_fine/dfine.py:87-124).  Here parameters, gradients, both Adam moments and the EMA copy
// live in flat buffers (the nn.Parameters are views), so the whole step is HBM-streaming work:
//   sqnorm_kernel   sum g^2 over all trainable parameters            (1 read)
//   adamw_ema_kernel per group: reads p, g, m, v, ema; writes p, m, v, ema, g := 0
// PyTorch semantics reproduced: clip coefficient min(1, max_norm / (norm + 1e-6)); decoupled weight
// decay p *= 1 - lr*wd; m, v moments; bias corrections 1 - beta^t; denom = sqrt(v)/sqrt(bc2) + eps.
#include "common.h"

namespace dfine {

// Two deterministic stages (a fixed grid, fixed summation order): every data-parallel rank must derive the SAME
// clip coefficient from the same all-reduced gradient, bit for bit - an atomics-based sum drifts the replicas
// apart by ulps per step.  partial[b] = block b's sum; sqnorm_final_kernel adds the partials in index order.
constexpr int kSqnormBlocks = 4096;
__global__ __launch_bounds__(256) void sqnorm_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t step = (int64_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += step) {
        const float4 v = *reinterpret_cast<const float4 *>(g + i);
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (i < n) for (int64_t j = i; j < n && j < i + 4; ++j) acc += g[j] * g[j];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float *__restrict__ partial, int nblocks, float grad_scale,
                                                           float *__restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * grad_scale * grad_scale;
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale, max_norm, ema_momentum;
};

__global__ __launch_bounds__(256) void adamw_ema_kernel(float *__restrict__ p, float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v,
                                                        float *__restrict__ ema, int64_t n,
                                                        const float *__restrict__ sqnorm, AdamArgs a) {
    float clip = 1.f;
    if (a.max_norm > 0.f && sqnorm) {
        const float c = a.max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
        clip = c < 1.f ? c : 1.f;
    }
    const float gs = a.grad_scale * clip;
    const float decay = 1.f - a.lr * a.weight_decay;
    const float step_size = a.lr / a.bc1;
    auto upd = [&](float &pp, float gg, float &mm, float &vv, float &ee) {
        gg *= gs;
        pp *= decay;
        mm = a.beta1 * mm + (1.f - a.beta1) * gg;
        vv = a.beta2 * vv + (1.f - a.beta2) * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) / a.bc2_sqrt + a.eps);
        ee = ee * a.ema_momentum + (1.f - a.ema_momentum) * pp;
    };
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t step = (int64_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += step) {
        float4 P = *reinterpret_cast<float4 *>(p + i), G = *reinterpret_cast<float4 *>(g + i);
        float4 M = *reinterpret_cast<float4 *>(m + i), V = *reinterpret_cast<float4 *>(v + i);
        float4 E = ema ? *reinterpret_cast<float4 *>(ema + i) : make_float4(0, 0, 0, 0);
        upd(P.x, G.x, M.x, V.x, E.x); upd(P.y, G.y, M.y, V.y, E.y);
        upd(P.z, G.z, M.z, V.z, E.z); upd(P.w, G.w, M.w, V.w, E.w);
        *reinterpret_cast<float4 *>(p + i) = P; *reinterpret_cast<float4 *>(m + i) = M;
        *reinterpret_cast<float4 *>(v + i) = V;
        if (ema) *reinterpret_cast<float4 *>(ema + i) = E;
        *reinterpret_cast<float4 *>(g + i) = make_float4(0, 0, 0, 0);
    }
    if (i < n)
        for (int64_t j = i; j < n && j < i + 4; ++j) {
            float e = ema ? ema[j] : 0.f;
            upd(p[j], g[j], m[j], v[j], e);
            if (ema) ema[j] = e;
            g[j] = 0.f;
        }
}

__global__ __launch_bounds__(256) void ema_kernel(float *__restrict__ ema, const float *__restrict__ src, int64_t n,
                                                  float momentum) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (; i < n; i += step) ema[i] = ema[i] * momentum + (1.f - momentum) * src[i];
}

// Gathers many separately allocated fp32 tensors (the per-parameter gradients autograd produced) into
// one flat buffer: table[i] = {src pointer, destination offset, element count}, one entry per <= 64 K
// element chunk, one block per entry.
struct CopyEntry { const float *src; int64_t dst_off; int64_t n; };

template <bool ADD>
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyEntry *__restrict__ table, float *__restrict__ dst) {
    const CopyEntry e = table[blockIdx.x];
    float *d = dst + e.dst_off;
    const bool aligned = (((uintptr_t)e.src | (uintptr_t)d) & 15) == 0;
    if (aligned) {
        const int64_t n4 = e.n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            float4 v = reinterpret_cast<const float4 *>(e.src)[i];
            if (ADD) {
                const float4 o = reinterpret_cast<float4 *>(d)[i];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            reinterpret_cast<float4 *>(d)[i] = v;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < e.n; i += 256) d[i] = ADD ? d[i] + e.src[i] : e.src[i];
    } else {
        for (int64_t i = threadIdx.x; i < e.n; i += 256) d[i] = ADD ? d[i] + e.src[i] : e.src[i];
    }
}

// bf16 shadow copies of many fp32 parameters with one launch: table[i] = {src pointer, dst (bf16) pointer, count}.
struct CastEntry { const float *src; uint16_t *dst; int64_t n; };

__global__ __launch_bounds__(256) void multi_cast_bf16_kernel(const CastEntry *__restrict__ table) {
    const CastEntry e = table[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < e.n; i += (int64_t)gridDim.x * 256)
        e.dst[i] = f32_to_bf16(e.src[i]);
}

static int grid_for(int64_t n, int per_thread) {
    int64_t b = (n / per_thread + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_grad_sqnorm_ws_floats(void) { return 1 + kSqnormBlocks; }

// out: dfine_grad_sqnorm_ws_floats() floats; out[0] = grad_scale^2 * sum(grad^2) (overwritten), the rest is scratch.
int dfine_grad_sqnorm(const float *grad, int64_t n, float grad_scale, float *out, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!grad || !out || n < 0) return DFINE_E_BADARG;
    int nb = grid_for(n, 4);
    if (nb > kSqnormBlocks) nb = kSqnormBlocks;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, grad, n, out + 1);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out + 1, nb, grad_scale, out);
    return check_launch();
}

int dfine_adamw_ema_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema, int64_t n,
                         const float *sqnorm, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, float grad_scale, float max_norm, float ema_momentum, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return DFINE_E_BADARG;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.grad_scale = grad_scale; a.max_norm = max_norm; a.ema_momentum = ema_momentum;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, ema, n, sqnorm, a);
    return check_launch();
}

// table: DEVICE array of n_entries {const float* src; int64 dst_off; int64 n} records (24 bytes each).
int dfine_multi_copy_f32(const void *table, int n_entries, float *dst, void *stream) {
    if (n_entries == 0) return DFINE_OK;
    if (!table || !dst || n_entries < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(multi_copy_kernel<false>, dim3(n_entries), dim3(256), 0, (hipStream_t)stream, (const CopyEntry *)table, dst);
    return check_launch();
}

// dst = src[0] + src[1] + ... + src[n - 1] (2 <= n <= 8 fp32 tensors of `count` elements, count % 4 == 0): the gradient of a
// token-stream tensor with several consumers in one pass - autograd adds them pairwise (n - 1 launches, 3 (n - 1) tensor passes
// against n + 1 here).  Left-to-right fp32 sum.
struct SumSrcs { const float *p[8]; };
__global__ __launch_bounds__(256) void sum_f32_kernel(SumSrcs srcs, int n, float *__restrict__ dst, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4 *>(srcs.p[0])[i];
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            if (k < n) {
                const float4 b = reinterpret_cast<const float4 *>(srcs.p[k])[i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
        }
        reinterpret_cast<float4 *>(dst)[i] = a;
    }
}

int dfine_sum_f32(const void *const *srcs, int n, float *dst, int64_t count, void *stream) {
    if (count == 0) return DFINE_OK;
    if (!srcs || !dst || n < 2 || n > 8 || count < 0 || count % 4) return DFINE_E_BADARG;
    SumSrcs s{};
    for (int k = 0; k < n; ++k) {
        if (!srcs[k] || ((uintptr_t)srcs[k] & 15)) return DFINE_E_BADARG;
        s.p[k] = (const float *)srcs[k];
    }
    if ((uintptr_t)dst & 15) return DFINE_E_BADARG;
    const int64_t nvec = count / 4;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, n, dst, nvec);
    return check_launch();
}

// Same table; dst[dst_offset + i] += src[i].  Records of one launch must not overlap in dst (one block per record, plain
// read-modify-write).
int dfine_multi_add_f32(const void *table, int n_entries, float *dst, void *stream) {
    if (n_entries == 0) return DFINE_OK;
    if (!table || !dst || n_entries < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(multi_copy_kernel<true>, dim3(n_entries), dim3(256), 0, (hipStream_t)stream, (const CopyEntry *)table, dst);
    return check_launch();
}

// table: DEVICE array of n_entries {const float* src; bf16* dst; int64 n} records (24 bytes each).
int dfine_multi_cast_bf16(const void *table, int n_entries, void *stream) {
    if (n_entries == 0) return DFINE_OK;
    if (!table || n_entries < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(multi_cast_bf16_kernel, dim3(16, n_entries), dim3(256), 0, (hipStream_t)stream, (const CastEntry *)table);
    return check_launch();
}

int dfine_ema_update(float *ema, const float *src, int64_t n, float momentum, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!ema || !src || n < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 1)), dim3(256), 0, (hipStream_t)stream, ema, src, n, momentum);
    return check_launch();
}

}  // extern "C"
