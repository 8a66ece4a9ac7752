// A10 / A15 - segmentation head (BASELINE configs[4]: D-FINE-x + mask head at 960 x 960) on gfx950:
//   * GroupNorm (+ ReLU) of the MaskDecoder (ref src/d_fine/arch/dfine_decoder.py:316-370: nn.GroupNorm(32, C) after the
//     lateral 1x1 convs and the two 3x3 convs), forward and backward;
//   * bilinear resize, align_corners = False (F.interpolate in MaskDecoder.forward :353-370 and in the criterion's target
//     preparation dfine_criterion.py:239-270), forward (optionally adding onto the destination: the lateral upsample-sum) and
//     backward as a GATHER over the output pixels that read an input pixel (no atomics);
//   * cropped BCE + Dice of the matched masks (dfine_criterion.py:335-450,504-556): one pass for the per-mask sums, one for
//     the gradient, straight on the [B, Q, H, W] logits through the (image, query) plan - no gathered copy;
//   * the matcher's pairwise mask costs (matcher.py:19-71,175-237): sum_p sigmoid(x_qp) g_tp and sum_p (pos - neg)(x_qp) g_tp
//     for every (query, target) of an image in one pass over the image's mask logits (the reference: four [Q, HW] x [HW, T]
//     matmuls on materialised sigmoid / focal maps).
// All of it is HBM-bound element-wise / reduction work on maps of B x 256 x (H/4)^2 elements (118 M at 960 x 960, bs 8):
// 16-byte accesses, fp32 arithmetic, one (image, channel) plane per workgroup where a reduction is needed.
// The dense contractions of the head - the lateral 1x1 / fusion 3x3 convolutions and the mask-logit einsum
// `bqc,bchw->bqhw` (:925-932) - run on the MFMA convolution kernels of conv.hip (per-image weights for the einsum).
#include "common.h"

namespace dfine {

constexpr int kMaskThreads = 256;

__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm.  x, y [B, C, HW]; part [B, C, 2] per-plane sums; stat [B, G, 2] = (mean, rstd).
template <typename T>
__global__ __launch_bounds__(kMaskThreads) void gn_plane_sums_kernel(const T *__restrict__ x, float *__restrict__ part, int HW) {
    __shared__ float red[8];
    const T *p = x + (int64_t)blockIdx.x * HW;
    float s = 0.f, q = 0.f;
    const int n4 = HW & ~3;
    for (int i = threadIdx.x * 4; i < n4; i += kMaskThreads * 4) {
        const f32x4 v = Vec4<T>::load(p + i);
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (int i = n4 + threadIdx.x; i < HW; i += kMaskThreads) { const float v = load_f(p + i); s += v; q += v * v; }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) { part[(int64_t)blockIdx.x * 2] = s; part[(int64_t)blockIdx.x * 2 + 1] = q; }
}

__global__ void gn_finalize_kernel(const float *__restrict__ part, float *__restrict__ stat, int BG, int cpg, int HW, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BG) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < cpg; ++c) { s += part[((int64_t)i * cpg + c) * 2]; q += part[((int64_t)i * cpg + c) * 2 + 1]; }
    const double n = (double)cpg * HW, mean = s / n;
    const double var = fmax(q / n - mean * mean, 0.0);
    stat[(int64_t)i * 2] = (float)mean;
    stat[(int64_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

template <typename T>
__global__ __launch_bounds__(kMaskThreads) void gn_apply_kernel(const T *__restrict__ x, T *__restrict__ y, const float *__restrict__ stat,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                int C, int cpg, int HW, int relu) {
    const int plane = blockIdx.y, c = plane % C, bg = plane / cpg;           // plane = b * C + c; (b, group) = plane / cpg
    const float mean = stat[(int64_t)bg * 2], rstd = stat[(int64_t)bg * 2 + 1];
    const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
    const T *p = x + (int64_t)plane * HW;
    T *o = y + (int64_t)plane * HW;
    const int n4 = HW & ~3;
    for (int i = (blockIdx.x * kMaskThreads + threadIdx.x) * 4; i < n4; i += gridDim.x * kMaskThreads * 4) {
        f32x4 v = Vec4<T>::load(p + i);
        v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        Vec4<T>::store(o + i, v);
    }
    if (blockIdx.x == 0)
        for (int i = n4 + threadIdx.x; i < HW; i += kMaskThreads) {
            float v = load_f(p + i) * sc + sh;
            store_f(o + i, relu ? fmaxf(v, 0.f) : v);
        }
}

// backward: dz = dy * [y > 0] (ReLU folded in); per plane S1 = sum dz, S2 = sum dz * xhat
template <typename T>
__global__ __launch_bounds__(kMaskThreads) void gn_bwd_sums_kernel(const T *__restrict__ x, const T *__restrict__ dy, const float *__restrict__ stat,
                                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                   float *__restrict__ part, int C, int cpg, int HW, int relu) {
    __shared__ float red[8];
    const int plane = blockIdx.x, c = plane % C, bg = plane / cpg;
    const float mean = stat[(int64_t)bg * 2], rstd = stat[(int64_t)bg * 2 + 1];
    const float g = gamma[c], b = beta[c];
    const T *p = x + (int64_t)plane * HW, *d = dy + (int64_t)plane * HW;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += kMaskThreads) {
        const float xh = (load_f(p + i) - mean) * rstd;
        float dz = load_f(d + i);
        if (relu && xh * g + b <= 0.f) dz = 0.f;
        s1 += dz; s2 += dz * xh;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { part[(int64_t)plane * 2] = s1; part[(int64_t)plane * 2 + 1] = s2; }
}

// gstat [B, G, 2] = (mean over the group of gamma * dz, mean of gamma * dz * xhat)
__global__ void gn_bwd_finalize_kernel(const float *__restrict__ part, const float *__restrict__ gamma, float *__restrict__ gstat, int BG,
                                       int G, int cpg, int HW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BG) return;
    const int g = i % G;
    double a = 0.0, b = 0.0;
    for (int c = 0; c < cpg; ++c) {
        const double gm = gamma[g * cpg + c];
        a += gm * part[((int64_t)i * cpg + c) * 2];
        b += gm * part[((int64_t)i * cpg + c) * 2 + 1];
    }
    const double n = (double)cpg * HW;
    gstat[(int64_t)i * 2] = (float)(a / n);
    gstat[(int64_t)i * 2 + 1] = (float)(b / n);
}

template <typename T>
__global__ __launch_bounds__(kMaskThreads) void gn_bwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ dx,
                                                                    const float *__restrict__ stat, const float *__restrict__ gstat,
                                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                    int C, int cpg, int HW, int relu) {
    const int plane = blockIdx.y, c = plane % C, bg = plane / cpg;
    const float mean = stat[(int64_t)bg * 2], rstd = stat[(int64_t)bg * 2 + 1];
    const float m1 = gstat[(int64_t)bg * 2], m2 = gstat[(int64_t)bg * 2 + 1];
    const float g = gamma[c], b = beta[c];
    const T *p = x + (int64_t)plane * HW, *d = dy + (int64_t)plane * HW;
    T *o = dx + (int64_t)plane * HW;
    for (int i = blockIdx.x * kMaskThreads + threadIdx.x; i < HW; i += gridDim.x * kMaskThreads) {
        const float xh = (load_f(p + i) - mean) * rstd;
        float dz = load_f(d + i);
        if (relu && xh * g + b <= 0.f) dz = 0.f;
        store_f(o + i, rstd * (g * dz - m1 - xh * m2));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Bilinear resize, align_corners = False (ATen upsample_bilinear2d: src = max((dst + 0.5) * in / out - 0.5, 0)).
__device__ __forceinline__ void bil_src(int d, float scale, int in_size, int *i0, int *i1, float *f) {
    float s = ((float)d + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    const int a = (int)s;                               // s >= 0: truncation = floor
    *i0 = min(a, in_size - 1);
    *i1 = min(a + 1, in_size - 1);
    *f = s - (float)a;
}

template <typename T>
__global__ __launch_bounds__(kMaskThreads) void bilinear_fwd_kernel(const T *__restrict__ x, const T *base, T *y, int Hi, int Wi, int Ho, int Wo,
                                                                    float sy, float sx) {
    const int plane = blockIdx.y;
    const T *p = x + (int64_t)plane * Hi * Wi;
    T *o = y + (int64_t)plane * Ho * Wo;
    const T *ob = base ? base + (int64_t)plane * Ho * Wo : nullptr;
    for (int i = blockIdx.x * kMaskThreads + threadIdx.x; i < Ho * Wo; i += gridDim.x * kMaskThreads) {
        const int oy = i / Wo, ox = i - oy * Wo;
        int y0, y1, x0, x1; float fy, fx;
        bil_src(oy, sy, Hi, &y0, &y1, &fy);
        bil_src(ox, sx, Wi, &x0, &x1, &fx);
        const float v00 = load_f(p + y0 * Wi + x0), v01 = load_f(p + y0 * Wi + x1);
        const float v10 = load_f(p + y1 * Wi + x0), v11 = load_f(p + y1 * Wi + x1);
        float v = (1.f - fy) * ((1.f - fx) * v00 + fx * v01) + fy * ((1.f - fx) * v10 + fx * v11);
        if (ob) v += load_f(ob + i);
        store_f(o + i, v);
    }
}

// dx[iy, ix] = sum over the output pixels whose two taps per axis include (iy, ix): candidates come from inverting the source
// map with one pixel of slack on both sides, each candidate's weight from the forward arithmetic itself.
template <typename T>
__global__ __launch_bounds__(kMaskThreads) void bilinear_bwd_kernel(const T *__restrict__ dy, T *__restrict__ dx, int Hi, int Wi, int Ho, int Wo,
                                                                    float sy, float sx) {
    const int plane = blockIdx.y;
    const T *d = dy + (int64_t)plane * Ho * Wo;
    T *o = dx + (int64_t)plane * Hi * Wi;
    const float ry = 1.f / sy, rx = 1.f / sx;             // out / in
    for (int i = blockIdx.x * kMaskThreads + threadIdx.x; i < Hi * Wi; i += gridDim.x * kMaskThreads) {
        const int iy = i / Wi, ix = i - iy * Wi;
        const int oy_lo = max((int)floorf(((float)iy - 1.f + 0.5f) * ry - 0.5f) - 1, 0);
        const int oy_hi = iy == Hi - 1 ? Ho - 1 : min((int)ceilf(((float)iy + 1.f + 0.5f) * ry - 0.5f) + 1, Ho - 1);
        const int ox_lo = max((int)floorf(((float)ix - 1.f + 0.5f) * rx - 0.5f) - 1, 0);
        const int ox_hi = ix == Wi - 1 ? Wo - 1 : min((int)ceilf(((float)ix + 1.f + 0.5f) * rx - 0.5f) + 1, Wo - 1);
        float acc = 0.f;
        for (int oy = (iy == 0 ? 0 : oy_lo); oy <= oy_hi; ++oy) {
            int y0, y1; float fy;
            bil_src(oy, sy, Hi, &y0, &y1, &fy);
            const float wy = (y0 == iy ? 1.f - fy : 0.f) + (y1 == iy ? fy : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = (ix == 0 ? 0 : ox_lo); ox <= ox_hi; ++ox) {
                int x0, x1; float fx;
                bil_src(ox, sx, Wi, &x0, &x1, &fx);
                const float wx = (x0 == ix ? 1.f - fx : 0.f) + (x1 == ix ? fx : 0.f);
                if (wx != 0.f) row += wx * load_f(d + oy * Wo + ox);
            }
            acc += wy * row;
        }
        store_f(o + i, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Cropped BCE + Dice of M matched masks.  pm [B, Q, HW] logits; plan_b / plan_q [M]; tgt [M, HW] f32 in [0, 1];
// boxes [M, 4] f32 (x1, y1, x2, y2 in mask pixels).  sums [M, 4] = (bce inside, p * t inside, p inside, t inside).
__device__ __forceinline__ bool box_has(const float4 b, int x, int y) {
    const float fx = (float)x, fy = (float)y;
    return fx >= b.x && fx < b.z && fy >= b.y && fy < b.w;
}

template <typename T>
__global__ __launch_bounds__(kMaskThreads) void mask_loss_sums_kernel(const T *__restrict__ pm, const int64_t *__restrict__ plan_b,
                                                                      const int64_t *__restrict__ plan_q, const int64_t *__restrict__ plan_t,
                                                                      const float *__restrict__ tgt, const float *__restrict__ boxes,
                                                                      float *__restrict__ sums, int Q, int H, int W) {
    __shared__ float red[8];
    const int m = blockIdx.x, HW = H * W;
    const T *p = pm + ((int64_t)plan_b[m] * Q + plan_q[m]) * HW;
    const int64_t row = plan_t ? plan_t[m] : m;           // row of tgt / boxes: the matched target, or m when they are gathered
    const float *t = tgt + row * HW;
    const float4 bx = *reinterpret_cast<const float4 *>(boxes + row * 4);
    // only the rows / columns the box can cover are visited
    const int y_lo = max((int)ceilf(bx.y), 0), y_hi = min((int)ceilf(bx.w), H);
    const int x_lo = max((int)ceilf(bx.x), 0), x_hi = min((int)ceilf(bx.z), W);
    const int bw = max(x_hi - x_lo, 0), n = bw * max(y_hi - y_lo, 0);
    float s_bce = 0.f, s_pt = 0.f, s_p = 0.f, s_t = 0.f;
    for (int i = threadIdx.x; i < n; i += kMaskThreads) {
        const int y = y_lo + i / bw, x = x_lo + i % bw;
        if (!box_has(bx, x, y)) continue;
        const float z = load_f(p + y * W + x), g = t[y * W + x];
        // binary_cross_entropy_with_logits: max(z, 0) - z g + log(1 + exp(-|z|))
        s_bce += fmaxf(z, 0.f) - z * g + log1pf(__expf(-fabsf(z)));
        const float pr = 1.f / (1.f + __expf(-z));
        s_pt += pr * g; s_p += pr; s_t += g;
    }
    s_bce = block_sum(s_bce, red); s_pt = block_sum(s_pt, red); s_p = block_sum(s_p, red); s_t = block_sum(s_t, red);
    if (threadIdx.x == 0) *reinterpret_cast<float4 *>(sums + (int64_t)m * 4) = make_float4(s_bce, s_pt, s_p, s_t);
}

// grad [B, Q, HW] (zero-filled by the caller; the matched planes are overwritten): d(loss) / d(logit) with
// loss = g_bce * mean_m(bce_m / area_m) + g_dice * mean_m(1 - (2 pt_m + eps) / (p_m + t_m + eps)); coef [M, 3] =
// (g_bce / (M area_m), -g_dice 2 / (M den_m), g_dice (2 pt_m + eps) / (M den_m^2)) computed by the host wrapper from `sums`.
template <typename T>
__global__ __launch_bounds__(kMaskThreads) void mask_loss_grad_kernel(const T *__restrict__ pm, const int64_t *__restrict__ plan_b,
                                                                      const int64_t *__restrict__ plan_q, const int64_t *__restrict__ plan_t,
                                                                      const float *__restrict__ tgt, const float *__restrict__ boxes,
                                                                      const float *__restrict__ coef, T *__restrict__ grad, int Q, int H, int W) {
    const int m = blockIdx.y, HW = H * W;
    const int64_t plane = ((int64_t)plan_b[m] * Q + plan_q[m]) * HW;
    const T *p = pm + plane;
    T *g_out = grad + plane;
    const int64_t row = plan_t ? plan_t[m] : m;
    const float *t = tgt + row * HW;
    const float4 bx = *reinterpret_cast<const float4 *>(boxes + row * 4);
    const float c_bce = coef[(int64_t)m * 3], c_pt = coef[(int64_t)m * 3 + 1], c_p = coef[(int64_t)m * 3 + 2];
    for (int i = blockIdx.x * kMaskThreads + threadIdx.x; i < HW; i += gridDim.x * kMaskThreads) {
        const int y = i / W, x = i - y * W;
        float gr = 0.f;
        if (box_has(bx, x, y)) {
            const float z = load_f(p + i), g = t[i];
            const float pr = 1.f / (1.f + __expf(-z));
            gr = c_bce * (pr - g) + (c_pt * g + c_p) * pr * (1.f - pr);
        }
        store_f(g_out + i, gr);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Matcher mask costs.  pm [B, Qall, HW] logits (the LAST Q queries of every image are matched); gt [sumT, HW] f32 (the
// images' target masks at mask resolution, concatenated); toff [B + 1].  out [B, Q, Tmax, 2] = (sum_p sigmoid(x) g,
// sum_p (pos - neg)(x) g); qsum [B, Q, 2] = (sum_p sigmoid(x), sum_p neg(x)).
// block = 8 waves = 8 queries of one image; the targets' pixels go through LDS in chunks of kMcPix shared by the 8 queries.
constexpr int kMcPix = 256, kMcT = 32;

// one block of NG * 8 targets (compile-time count: the accumulators stay in registers, groups past the image's target
// count cost nothing) against the wave's query over all pixels
template <typename T, int NG>
__device__ __forceinline__ void mask_cost_block(const T *__restrict__ p, const float *__restrict__ gt, float (*s_g)[kMcPix], int t_first,
                                                int tn, int HW, float alpha, float gamma, bool first, float &s_sig, float &s_neg,
                                                float *__restrict__ o2, bool live, int lane) {
    float a_d[NG * 8], a_f[NG * 8];
#pragma unroll
    for (int t = 0; t < NG * 8; ++t) { a_d[t] = 0.f; a_f[t] = 0.f; }
    // software pipeline over the 256-pixel chunks: the next chunk's target pixels (staged through LDS for the 8 queries of the
    // block) and this wave's 4 x 64 logits are loaded into registers BEFORE the current chunk's arithmetic.  (Loaded and consumed
    // in the same chunk, the 225 chunks of a 240 x 240 mask were 225 memory round trips in a row: 1.8 ms per launch, config #5.)
    constexpr int NGF = NG * 8 * kMcPix / 512;                     // staged target values per thread and chunk
    float gpf[NGF];
    T ppf[kMcPix / 64], pcur[kMcPix / 64];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int j = 0; j < NGF; ++j) {
            const int i = threadIdx.x + 512 * j;
            const int t = i / kMcPix, px = i - t * kMcPix;
            gpf[j] = (t < tn && p0 + px < HW) ? gt[(int64_t)(t_first + t) * HW + p0 + px] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kMcPix / 64; ++k) {
            const int px = p0 + k * 64 + lane;
            ppf[k] = px < HW ? p[px] : T(0);
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < HW; p0 += kMcPix) {
        __syncthreads();                                           // every wave is done with the previous chunk's targets
#pragma unroll
        for (int j = 0; j < NGF; ++j) {
            const int i = threadIdx.x + 512 * j;
            s_g[i / kMcPix][i % kMcPix] = gpf[j];
        }
#pragma unroll
        for (int k = 0; k < kMcPix / 64; ++k) pcur[k] = ppf[k];
        __syncthreads();
        if (p0 + kMcPix < HW) fetch(p0 + kMcPix);                  // in flight during the arithmetic below
#pragma unroll
        for (int k = 0; k < kMcPix / 64; ++k) {
            const int px = k * 64 + lane;
            if (p0 + px < HW) {
                const float x = load_f(&pcur[k]);
                // hardware exp / log (1 ulp-class relative error, averaged over the H * W terms of a cost sum): the accurate
                // library forms made this kernel VALU-bound at 2.3 ms per head of config #5
                const float pr = __builtin_amdgcn_rcpf(1.f + __expf(-x));
                const float pg = gamma == 2.f ? pr * pr : __powf(pr, gamma);
                const float qg = gamma == 2.f ? (1.f - pr) * (1.f - pr) : __powf(1.f - pr, gamma);
                const float neg = (1.f - alpha) * pg * (-__logf(1.f - pr + 1e-8f));
                const float pos = alpha * qg * (-__logf(pr + 1e-8f));
                if (first) { s_sig += pr; s_neg += neg; }
                const float pn = pos - neg;
#pragma unroll
                for (int t = 0; t < NG * 8; ++t) {
                    const float g = s_g[t][px];
                    a_d[t] += pr * g; a_f[t] += pn * g;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NG * 8; ++t) {
        float d = a_d[t], f = a_f[t];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { d += __shfl_xor(d, o, 64); f += __shfl_xor(f, o, 64); }
        if (lane == 0 && live && t < tn) { o2[t * 2] = d; o2[t * 2 + 1] = f; }
    }
}

template <typename T>
__global__ __launch_bounds__(512) void mask_cost_kernel(const T *__restrict__ pm, const float *__restrict__ gt, const int *__restrict__ toff,
                                                        float *__restrict__ out, float *__restrict__ qsum, int Qall, int Q, int HW, int Tmax,
                                                        float alpha, float gamma) {
    __shared__ float s_g[kMcT][kMcPix];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 8 + wave;
    const int t0 = toff[b], nt = toff[b + 1] - t0;
    const bool live = q < Q;
    const T *p = pm + ((int64_t)b * Qall + (Qall - Q) + (live ? q : 0)) * HW;
    float s_sig = 0.f, s_neg = 0.f;
    for (int tb = 0; tb < max(nt, 1); tb += kMcT) {
        const int tn = max(min(kMcT, nt - tb), 0);
        float *o2 = out + (((int64_t)b * Q + (live ? q : 0)) * Tmax + tb) * 2;
        const int ng = (tn + 7) / 8;                                  // block-uniform
        if (ng <= 1) mask_cost_block<T, 1>(p, gt, s_g, t0 + tb, tn, HW, alpha, gamma, tb == 0, s_sig, s_neg, o2, live, lane);
        else if (ng == 2) mask_cost_block<T, 2>(p, gt, s_g, t0 + tb, tn, HW, alpha, gamma, tb == 0, s_sig, s_neg, o2, live, lane);
        else if (ng == 3) mask_cost_block<T, 3>(p, gt, s_g, t0 + tb, tn, HW, alpha, gamma, tb == 0, s_sig, s_neg, o2, live, lane);
        else mask_cost_block<T, 4>(p, gt, s_g, t0 + tb, tn, HW, alpha, gamma, tb == 0, s_sig, s_neg, o2, live, lane);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s_sig += __shfl_xor(s_sig, o, 64); s_neg += __shfl_xor(s_neg, o, 64); }
    if (lane == 0 && live) { qsum[((int64_t)b * Q + q) * 2] = s_sig; qsum[((int64_t)b * Q + q) * 2 + 1] = s_neg; }
}

static int plane_blocks(int HW, int per_thread) {
    int b = (HW + kMaskThreads * per_thread - 1) / (kMaskThreads * per_thread);
    return b < 1 ? 1 : (b > 64 ? 64 : b);
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_groupnorm_ws_floats(int B, int C, int G) { return (int64_t)B * C * 2 + (int64_t)B * G * 2; }

// y = [relu](GroupNorm_G(x) * gamma + beta); x, y [B, C, HW] dtype; stat [B, G, 2] f32 out (mean, rstd: saved for backward);
// ws: dfine_groupnorm_ws_floats(B, C, G) floats.
int dfine_groupnorm_fwd(const void *x, void *y, const float *gamma, const float *beta, float *stat, float *ws, int dtype, int B, int C,
                        int HW, int G, float eps, int relu, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !y || !gamma || !beta || !stat || !ws || C < 1 || G < 1 || C % G || HW < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int cpg = C / G;
    if (dtype == DFINE_F32) hipLaunchKernelGGL(gn_plane_sums_kernel<float>, dim3(B * C), dim3(kMaskThreads), 0, st, (const float *)x, ws, HW);
    else if (dtype == DFINE_BF16) hipLaunchKernelGGL(gn_plane_sums_kernel<uint16_t>, dim3(B * C), dim3(kMaskThreads), 0, st, (const uint16_t *)x, ws, HW);
    else return DFINE_E_BADARG;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 63) / 64), dim3(64), 0, st, ws, stat, B * G, cpg, HW, eps);
    const dim3 grid(plane_blocks(HW, 8), B * C);
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(kMaskThreads), 0, st, (const float *)x, (float *)y, stat, gamma, beta, C, cpg, HW, relu);
    else
        hipLaunchKernelGGL(gn_apply_kernel<uint16_t>, grid, dim3(kMaskThreads), 0, st, (const uint16_t *)x, (uint16_t *)y, stat, gamma, beta, C, cpg, HW, relu);
    return check_launch();
}

// dx [B, C, HW] dtype; part [B, C, 2] f32 out: per-plane (sum dz, sum dz xhat) - d(gamma)[c] = sum_b part[b, c, 1],
// d(beta)[c] = sum_b part[b, c, 0]; ws: B * G * 2 floats.
int dfine_groupnorm_bwd(const void *x, const void *dy, void *dx, const float *gamma, const float *beta, const float *stat, float *part,
                        float *ws, int dtype, int B, int C, int HW, int G, int relu, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !dy || !dx || !gamma || !beta || !stat || !part || !ws || C < 1 || G < 1 || C % G || HW < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int cpg = C / G;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(gn_bwd_sums_kernel<float>, dim3(B * C), dim3(kMaskThreads), 0, st, (const float *)x, (const float *)dy, stat, gamma, beta, part, C, cpg, HW, relu);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(gn_bwd_sums_kernel<uint16_t>, dim3(B * C), dim3(kMaskThreads), 0, st, (const uint16_t *)x, (const uint16_t *)dy, stat, gamma, beta, part, C, cpg, HW, relu);
    else return DFINE_E_BADARG;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((B * G + 63) / 64), dim3(64), 0, st, part, gamma, ws, B * G, G, cpg, HW);
    const dim3 grid(plane_blocks(HW, 4), B * C);
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(kMaskThreads), 0, st, (const float *)x, (const float *)dy, (float *)dx, stat, ws, gamma, beta, C, cpg, HW, relu);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<uint16_t>, grid, dim3(kMaskThreads), 0, st, (const uint16_t *)x, (const uint16_t *)dy, (uint16_t *)dx, stat, ws, gamma, beta, C, cpg, HW, relu);
    return check_launch();
}

// y [planes, Ho, Wo] = [base +] bilinear(x [planes, Hi, Wi]), align_corners = False; base may be NULL or equal y
int dfine_bilinear_fwd(const void *x, const void *base, void *y, int dtype, int planes, int Hi, int Wi, int Ho, int Wo, void *stream) {
    if (planes == 0) return DFINE_OK;
    if (!x || !y || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return DFINE_E_BADARG;
    const dim3 grid(plane_blocks(Ho * Wo, 2), planes);
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(bilinear_fwd_kernel<float>, grid, dim3(kMaskThreads), 0, st, (const float *)x, (const float *)base, (float *)y, Hi, Wi, Ho, Wo, sy, sx);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(bilinear_fwd_kernel<uint16_t>, grid, dim3(kMaskThreads), 0, st, (const uint16_t *)x, (const uint16_t *)base, (uint16_t *)y, Hi, Wi, Ho, Wo, sy, sx);
    else return DFINE_E_BADARG;
    return check_launch();
}

// dx [planes, Hi, Wi] = adjoint of dfine_bilinear_fwd applied to dy [planes, Ho, Wo]
int dfine_bilinear_bwd(const void *dy, void *dx, int dtype, int planes, int Hi, int Wi, int Ho, int Wo, void *stream) {
    if (planes == 0) return DFINE_OK;
    if (!dy || !dx || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return DFINE_E_BADARG;
    const dim3 grid(plane_blocks(Hi * Wi, 1), planes);
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(bilinear_bwd_kernel<float>, grid, dim3(kMaskThreads), 0, st, (const float *)dy, (float *)dx, Hi, Wi, Ho, Wo, sy, sx);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(bilinear_bwd_kernel<uint16_t>, grid, dim3(kMaskThreads), 0, st, (const uint16_t *)dy, (uint16_t *)dx, Hi, Wi, Ho, Wo, sy, sx);
    else return DFINE_E_BADARG;
    return check_launch();
}

int dfine_mask_loss_sums(const void *pm, const int64_t *plan_b, const int64_t *plan_q, const int64_t *plan_t, const float *tgt,
                         const float *boxes, float *sums, int dtype, int M, int Q, int H, int W, void *stream) {
    if (M == 0) return DFINE_OK;
    if (!pm || !plan_b || !plan_q || !tgt || !boxes || !sums || Q < 1 || H < 1 || W < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(mask_loss_sums_kernel<float>, dim3(M), dim3(kMaskThreads), 0, st, (const float *)pm, plan_b, plan_q, plan_t, tgt, boxes, sums, Q, H, W);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(mask_loss_sums_kernel<uint16_t>, dim3(M), dim3(kMaskThreads), 0, st, (const uint16_t *)pm, plan_b, plan_q, plan_t, tgt, boxes, sums, Q, H, W);
    else return DFINE_E_BADARG;
    return check_launch();
}

int dfine_mask_loss_grad(const void *pm, const int64_t *plan_b, const int64_t *plan_q, const int64_t *plan_t, const float *tgt,
                         const float *boxes, const float *coef, void *grad, int dtype, int M, int Q, int H, int W, void *stream) {
    if (M == 0) return DFINE_OK;
    if (!pm || !plan_b || !plan_q || !tgt || !boxes || !coef || !grad || Q < 1 || H < 1 || W < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(plane_blocks(H * W, 4), M);
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(mask_loss_grad_kernel<float>, grid, dim3(kMaskThreads), 0, st, (const float *)pm, plan_b, plan_q, plan_t, tgt, boxes, coef, (float *)grad, Q, H, W);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(mask_loss_grad_kernel<uint16_t>, grid, dim3(kMaskThreads), 0, st, (const uint16_t *)pm, plan_b, plan_q, plan_t, tgt, boxes, coef, (uint16_t *)grad, Q, H, W);
    else return DFINE_E_BADARG;
    return check_launch();
}

int dfine_mask_cost(const void *pm, const float *gt, const int *toff, float *out, float *qsum, int dtype, int B, int Qall, int Q, int HW,
                    int Tmax, float alpha, float gamma, void *stream) {
    if (B == 0 || Tmax == 0) return DFINE_OK;
    if (!pm || !gt || !toff || !out || !qsum || Q < 1 || Qall < Q || HW < 1 || Tmax < 0) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((Q + 7) / 8, B);
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(mask_cost_kernel<float>, grid, dim3(512), 0, st, (const float *)pm, gt, toff, out, qsum, Qall, Q, HW, Tmax, alpha, gamma);
    else if (dtype == DFINE_BF16)
        hipLaunchKernelGGL(mask_cost_kernel<uint16_t>, grid, dim3(512), 0, st, (const uint16_t *)pm, gt, toff, out, qsum, Qall, Q, HW, Tmax, alpha, gamma);
    else return DFINE_E_BADARG;
    return check_launch();
}

}  // extern "C"
