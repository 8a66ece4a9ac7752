// A7 - multi-scale deformable attention gather, forward + backward, for gfx950.
//
// Reference semantics: deformable_attention_core_func_v2 (src/d_fine/arch/utils.py:191-264) =
// per level F.grid_sample(value_l, 2*loc-1, bilinear, padding zeros, align_corners=False), i.e.
// pixel coordinate = loc * size - 0.5, followed by sum_p weight_p * sample_p; the FUSED
// variant also does MSDeformableAttention.forward's softmax over the P points and
// loc = ref_xy + offset * (1/points_of_level) * ref_wh * offset_scale
// (src/d_fine/arch/dfine_decoder.py:147,156-166).
//
// Data layout / mapping (HBM-bound gather, no MFMA):
//  * value stays [B, L, H, D] - the encoder memory itself; the D channels of one (pixel, head)
//    are one contiguous 64 B (bf16) / 128 B (f32) segment, read by D/4 adjacent lanes with one
//    8 B / 16 B load each.
//  * a "task" is one (b, q, head); D/4 lanes own a task and walk its P points x 4 corners with
//    fp32 accumulation in registers; a 256-thread block owns 1024/D consecutive (b,q) of ONE
//    head.
//  * head = blockIdx % H: with H = 8 the dispatcher's round-robin (block b -> XCD b % 8) pins
//    one head per XCD, so each XCD's 4 MiB L2 only ever sees its own 1/8 slice of `value`
//    (0.5 MB bf16 per image) - placement affects speed only, never results.
//  * sampling coordinates / weights of the block's tasks are staged once through LDS with
//    coalesced loads (fused mode: raw offsets + logits, softmax done from LDS).
//  * backward: same walk; d(value) by hardware f32 atomics into an f32 accumulator (zeroed by the
//    caller), d(loc)/d(weight) reduced over the task's lanes with wave shuffles.
#include <cstdlib>

#include "common.h"

namespace dfine {

struct MsdaLevels {
    int n_points;                       // P
    int start[DFINE_MAX_POINTS];        // first value row of the point's level
    int h[DFINE_MAX_POINTS];
    int w[DFINE_MAX_POINTS];
    float inv_n[DFINE_MAX_POINTS];      // 1 / (points of that level)
};

constexpr int kThreads = 256;

// -------------------------------------------------------------------------------------------
// stage (x, y, w) of the block's tasks into LDS.  s_x/s_y/s_w are [QPB][P].
// FUSED: x,y = sampling location from ref box + offsets; w = raw logit (softmax later).
template <typename T, bool FUSED>
__device__ __forceinline__ void stage_points(const MsdaLevels &lv, const float *__restrict__ loc,
                                             const float *__restrict__ weight,
                                             const float *__restrict__ ref,
                                             const T *__restrict__ offsets,
                                             const T *__restrict__ logits, float offset_scale,
                                             int head, int H, int q0, int nq, int total_q,
                                             float *s_x, float *s_y, float *s_w) {
    const int P = lv.n_points;
    for (int i = threadIdx.x; i < nq * P; i += kThreads) {
        const int qi = i / P, p = i - qi * P;
        const int64_t task = (int64_t)(q0 + qi) * H + head;
        if (FUSED) {
            const float4 r = *reinterpret_cast<const float4 *>(ref + (int64_t)(q0 + qi) * 4);
            const float ox = load_f(offsets + (task * P + p) * 2);
            const float oy = load_f(offsets + (task * P + p) * 2 + 1);
            // same association as the reference: ((off * scale) * ref_wh) * offset_scale
            s_x[i] = r.x + ox * lv.inv_n[p] * r.z * offset_scale;
            s_y[i] = r.y + oy * lv.inv_n[p] * r.w * offset_scale;
            s_w[i] = load_f(logits + task * P + p);
        } else {
            s_x[i] = loc[(task * P + p) * 2];
            s_y[i] = loc[(task * P + p) * 2 + 1];
            s_w[i] = weight[task * P + p];
        }
    }
}

struct Corner {
    int64_t row[4];   // value row (clamped) of the 4 corners, order (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    float bw[4];      // bilinear weight, 0 when the corner is outside the map
    float wx0, wx1, wy0, wy1;
    float ok[4];
};

__device__ __forceinline__ Corner corners(float lx, float ly, int start, int h, int w) {
    Corner c;
    const float x = lx * (float)w - 0.5f, y = ly * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    // clamp before the int conversion so wild coordinates (inf/NaN offsets) stay defined
    const int x0 = (int)fminf(fmaxf(xf, -2.f), (float)w + 1.f);
    const int y0 = (int)fminf(fmaxf(yf, -2.f), (float)h + 1.f);
    const int x1 = x0 + 1, y1 = y0 + 1;
    c.wx0 = 1.f - fx; c.wx1 = fx; c.wy0 = 1.f - fy; c.wy1 = fy;
    const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w;
    const bool vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
    const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
    c.row[0] = start + cy0 * w + cx0; c.row[1] = start + cy0 * w + cx1;
    c.row[2] = start + cy1 * w + cx0; c.row[3] = start + cy1 * w + cx1;
    c.ok[0] = (vy0 && vx0) ? 1.f : 0.f; c.ok[1] = (vy0 && vx1) ? 1.f : 0.f;
    c.ok[2] = (vy1 && vx0) ? 1.f : 0.f; c.ok[3] = (vy1 && vx1) ? 1.f : 0.f;
    c.bw[0] = c.wy0 * c.wx0 * c.ok[0]; c.bw[1] = c.wy0 * c.wx1 * c.ok[1];
    c.bw[2] = c.wy1 * c.wx0 * c.ok[2]; c.bw[3] = c.wy1 * c.wx1 * c.ok[3];
    return c;
}

// -------------------------------------------------------------------------------------------
template <typename T, int D, bool FUSED>
__global__ __launch_bounds__(kThreads) void msda_fwd_kernel(
    const T *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ weight,
    const float *__restrict__ ref, const T *__restrict__ offsets, const T *__restrict__ logits,
    T *__restrict__ out, MsdaLevels lv, int L, int H, int Lq, int total_q, float offset_scale) {
    constexpr int LPT = D / 4;              // lanes per task
    constexpr int QPB = kThreads / LPT;     // (b,q) pairs per block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = lv.n_points;
    float *s_x = smem, *s_y = smem + QPB * P, *s_w = smem + 2 * QPB * P;

    const int head = blockIdx.x % H;
    const int q0 = (blockIdx.x / H) * QPB;
    const int nq = min(QPB, total_q - q0);
    stage_points<T, FUSED>(lv, loc, weight, ref, offsets, logits, offset_scale, head, H, q0, nq,
                           total_q, s_x, s_y, s_w);
    __syncthreads();

    const int qi = threadIdx.x / LPT, c4 = (threadIdx.x % LPT) * 4;
    if (qi >= nq) return;
    const int gq = q0 + qi;                 // = b * Lq + q
    const int b = gq / Lq;
    const T *vbase = value + ((int64_t)b * L * H + head) * D + c4;
    const float *px = s_x + qi * P, *py = s_y + qi * P, *pw = s_w + qi * P;

    float wmax = 0.f, winv = 1.f;
    if (FUSED) {
        wmax = pw[0];
        for (int p = 1; p < P; ++p) wmax = fmaxf(wmax, pw[p]);
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += __expf(pw[p] - wmax);
        winv = 1.f / s;
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = 0; p < P; ++p) {
        const Corner c = corners(px[p], py[p], lv.start[p], lv.h[p], lv.w[p]);
        const float aw = FUSED ? __expf(pw[p] - wmax) * winv : pw[p];
        const int64_t stride = (int64_t)H * D;
        const f32x4 v0 = Vec4<T>::load(vbase + c.row[0] * stride);
        const f32x4 v1 = Vec4<T>::load(vbase + c.row[1] * stride);
        const f32x4 v2 = Vec4<T>::load(vbase + c.row[2] * stride);
        const f32x4 v3 = Vec4<T>::load(vbase + c.row[3] * stride);
        const float w0 = aw * c.bw[0], w1 = aw * c.bw[1], w2 = aw * c.bw[2], w3 = aw * c.bw[3];
        acc.x += w0 * v0.x + w1 * v1.x + w2 * v2.x + w3 * v3.x;
        acc.y += w0 * v0.y + w1 * v1.y + w2 * v2.y + w3 * v3.y;
        acc.z += w0 * v0.z + w1 * v1.z + w2 * v2.z + w3 * v3.z;
        acc.w += w0 * v0.w + w1 * v1.w + w2 * v2.w + w3 * v3.w;
    }
    Vec4<T>::store(out + ((int64_t)gq * H + head) * D + c4, acc);
}

// Forward, bf16, 8 channels (one 16-byte load) per lane: D/8 lanes per task, 2x fewer load
// instructions per gathered byte than the 4-channel mapping.
template <int D, bool FUSED>
__global__ __launch_bounds__(kThreads) void msda_fwd8_kernel(
    const uint16_t *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ weight,
    const float *__restrict__ ref, const uint16_t *__restrict__ offsets, const uint16_t *__restrict__ logits,
    uint16_t *__restrict__ out, MsdaLevels lv, int L, int H, int Lq, int total_q, float offset_scale) {
    constexpr int LPT = D / 8;
    constexpr int QPB = kThreads / LPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = lv.n_points;
    float *s_x = smem, *s_y = smem + QPB * P, *s_w = smem + 2 * QPB * P;
    const int head = blockIdx.x % H;
    const int q0 = (blockIdx.x / H) * QPB;
    const int nq = min(QPB, total_q - q0);
    stage_points<uint16_t, FUSED>(lv, loc, weight, ref, offsets, logits, offset_scale, head, H, q0, nq,
                                  total_q, s_x, s_y, s_w);
    __syncthreads();
    const int qi = threadIdx.x / LPT, c8 = (threadIdx.x % LPT) * 8;
    if (qi >= nq) return;
    const int gq = q0 + qi;
    const int b = gq / Lq;
    const uint16_t *vbase = value + ((int64_t)b * L * H + head) * D + c8;
    const float *px = s_x + qi * P, *py = s_y + qi * P, *pw = s_w + qi * P;
    float wmax = 0.f, winv = 1.f;
    if (FUSED) {
        wmax = pw[0];
        for (int p = 1; p < P; ++p) wmax = fmaxf(wmax, pw[p]);
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += __expf(pw[p] - wmax);
        winv = 1.f / s;
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)H * D;
#pragma unroll 2
    for (int p = 0; p < P; ++p) {
        const Corner c = corners(px[p], py[p], lv.start[p], lv.h[p], lv.w[p]);
        const float aw = FUSED ? __expf(pw[p] - wmax) * winv : pw[p];
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4 *>(vbase + c.row[k] * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = aw * c.bw[k];
            const uint32_t u[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += w * __uint_as_float(u[j] << 16);
                acc[2 * j + 1] += w * __uint_as_float(u[j] & 0xffff0000u);
            }
        }
    }
    uint4 o;
    o.x = pack_bf16x2(acc[0], acc[1]);
    o.y = pack_bf16x2(acc[2], acc[3]);
    o.z = pack_bf16x2(acc[4], acc[5]);
    o.w = pack_bf16x2(acc[6], acc[7]);
    *reinterpret_cast<uint4 *>(out + ((int64_t)gq * H + head) * D + c8) = o;
}

// -------------------------------------------------------------------------------------------
template <int LPT> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = LPT / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <typename T, int D, bool FUSED>
__global__ __launch_bounds__(kThreads) void msda_bwd_kernel(
    const T *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ weight,
    const float *__restrict__ ref, const T *__restrict__ offsets, const T *__restrict__ logits,
    const T *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_weight, T *__restrict__ grad_offsets, T *__restrict__ grad_logits,
    MsdaLevels lv, int L, int H, int Lq, int total_q, float offset_scale) {
    constexpr int LPT = D / 4;
    constexpr int QPB = kThreads / LPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = lv.n_points;
    float *s_x = smem, *s_y = smem + QPB * P, *s_w = smem + 2 * QPB * P;
    float *s_gx = smem + 3 * QPB * P, *s_gy = smem + 4 * QPB * P, *s_gw = smem + 5 * QPB * P;
    float *s_dot = smem + 6 * QPB * P;      // [QPB] sum_p w_p * gw_p  (softmax backward)

    const int head = blockIdx.x % H;
    const int q0 = (blockIdx.x / H) * QPB;
    const int nq = min(QPB, total_q - q0);
    stage_points<T, FUSED>(lv, loc, weight, ref, offsets, logits, offset_scale, head, H, q0, nq,
                           total_q, s_x, s_y, s_w);
    __syncthreads();

    const int qi = threadIdx.x / LPT, sub = threadIdx.x % LPT, c4 = sub * 4;
    if (qi < nq) {
        const int gq = q0 + qi;
        const int b = gq / Lq;
        const int64_t stride = (int64_t)H * D;
        const int64_t base = ((int64_t)b * L * H + head) * D + c4;
        const T *vbase = value + base;
        float *gvbase = grad_value + base;
        float *px = s_x + qi * P, *py = s_y + qi * P, *pw = s_w + qi * P;
        const f32x4 go = Vec4<T>::load(grad_out + ((int64_t)gq * H + head) * D + c4);

        float wmax = 0.f, winv = 1.f;
        if (FUSED) {
            wmax = pw[0];
            for (int p = 1; p < P; ++p) wmax = fmaxf(wmax, pw[p]);
            float s = 0.f;
            for (int p = 0; p < P; ++p) s += __expf(pw[p] - wmax);
            winv = 1.f / s;
        }
        float dot_acc = 0.f;
        for (int p = 0; p < P; ++p) {
            const Corner c = corners(px[p], py[p], lv.start[p], lv.h[p], lv.w[p]);
            const float aw = FUSED ? __expf(pw[p] - wmax) * winv : pw[p];
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v = Vec4<T>::load(vbase + c.row[k] * stride);
                d[k] = (go.x * v.x + go.y * v.y + go.z * v.z + go.w * v.w) * c.ok[k];
                const float g = aw * c.bw[k];
                if (g != 0.f) {
                    float *dst = gvbase + c.row[k] * stride;
                    unsafeAtomicAdd(dst + 0, g * go.x);
                    unsafeAtomicAdd(dst + 1, g * go.y);
                    unsafeAtomicAdd(dst + 2, g * go.z);
                    unsafeAtomicAdd(dst + 3, g * go.w);
                }
            }
            // partial (over this lane's 4 channels) gradients wrt weight, x, y
            float gw = c.wy0 * c.wx0 * d[0] + c.wy0 * c.wx1 * d[1] + c.wy1 * c.wx0 * d[2] + c.wy1 * c.wx1 * d[3];
            float gx = (c.wy0 * (d[1] - d[0]) + c.wy1 * (d[3] - d[2])) * aw * (float)lv.w[p];
            float gy = (c.wx0 * (d[2] - d[0]) + c.wx1 * (d[3] - d[1])) * aw * (float)lv.h[p];
            gw = group_sum<LPT>(gw);
            gx = group_sum<LPT>(gx);
            gy = group_sum<LPT>(gy);
            dot_acc += aw * gw;
            if (sub == 0) {
                s_gx[qi * P + p] = gx; s_gy[qi * P + p] = gy; s_gw[qi * P + p] = gw;
                if (FUSED) pw[p] = aw;      // logits no longer needed: keep the softmax weight
            }
        }
        if (sub == 0) s_dot[qi] = dot_acc;
    }
    __syncthreads();

    // coalesced write-out of the per-point gradients
    for (int i = threadIdx.x; i < nq * P; i += kThreads) {
        const int qj = i / P, p = i - qj * P;
        const int64_t task = (int64_t)(q0 + qj) * H + head;
        if (FUSED) {
            const float4 r = *reinterpret_cast<const float4 *>(ref + (int64_t)(q0 + qj) * 4);
            const float sc = lv.inv_n[p] * offset_scale;
            store_f(grad_offsets + (task * P + p) * 2, s_gx[i] * sc * r.z);
            store_f(grad_offsets + (task * P + p) * 2 + 1, s_gy[i] * sc * r.w);
            store_f(grad_logits + task * P + p, s_w[i] * (s_gw[i] - s_dot[qj]));
        } else {
            grad_loc[(task * P + p) * 2] = s_gx[i];
            grad_loc[(task * P + p) * 2 + 1] = s_gy[i];
            grad_weight[task * P + p] = s_gw[i];
        }
    }
}

// Backward, "wide" mapping: D lanes per task, one channel per lane.  The f32 atomics of one
// (task, corner) then cover one contiguous 4*D-byte run (a single 128 B line for D = 32) instead of
// four quarter-filled lines - the L2 atomic units are paced by line operations, not by bytes.
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;

// MERGE: contributions to one value row are summed in registers first (see msda_bwd_pair_kernel).
template <typename T, int D, bool FUSED, bool MERGE>
__global__ __launch_bounds__(kThreads) void msda_bwd_wide_kernel(
    const T *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ weight,
    const float *__restrict__ ref, const T *__restrict__ offsets, const T *__restrict__ logits,
    const T *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_weight, T *__restrict__ grad_offsets, T *__restrict__ grad_logits,
    MsdaLevels lv, int L, int H, int Lq, int total_q, float offset_scale) {
    constexpr int LPT = D;
    constexpr int QPB = kThreads / LPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = lv.n_points;
    float *s_x = smem, *s_y = smem + QPB * P, *s_w = smem + 2 * QPB * P;
    float *s_gx = smem + 3 * QPB * P, *s_gy = smem + 4 * QPB * P, *s_gw = smem + 5 * QPB * P;
    float *s_dot = smem + 6 * QPB * P;

    const int head = blockIdx.x % H;
    const int q0 = (blockIdx.x / H) * QPB;
    const int nq = min(QPB, total_q - q0);
    stage_points<T, FUSED>(lv, loc, weight, ref, offsets, logits, offset_scale, head, H, q0, nq,
                           total_q, s_x, s_y, s_w);
    __syncthreads();

    const int qi = threadIdx.x / LPT, ch = threadIdx.x % LPT;
    if (qi < nq) {
        const int gq = q0 + qi;
        const int b = gq / Lq;
        const int64_t stride = (int64_t)H * D;
        const int64_t base = ((int64_t)b * L * H + head) * D + ch;
        const T *vbase = value + base;
        float *gvbase = grad_value + base;
        float *px = s_x + qi * P, *py = s_y + qi * P, *pw = s_w + qi * P;
        const float go = load_f(grad_out + ((int64_t)gq * H + head) * D + ch);

        float wmax = 0.f, winv = 1.f;
        if (FUSED) {
            wmax = pw[0];
            for (int p = 1; p < P; ++p) wmax = fmaxf(wmax, pw[p]);
            float s = 0.f;
            for (int p = 0; p < P; ++p) s += __expf(pw[p] - wmax);
            winv = 1.f / s;
        }
        float *gvbase0 = grad_value + (int64_t)head * D + ch;      // (row 0 of image 0); a pending row is kept as b * L + row
        const int brow = b * L;
        int prow[4] = {-1, -1, -1, -1};
        float pa[4] = {0.f, 0.f, 0.f, 0.f};
        float dot_acc = 0.f;
        for (int p = 0; p < P; ++p) {
            const Corner c = corners(px[p], py[p], lv.start[p], lv.h[p], lv.w[p]);
            const float aw = FUSED ? __expf(pw[p] - wmax) * winv : pw[p];
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = go * load_f(vbase + c.row[k] * stride) * c.ok[k];
                const float g = aw * c.bw[k];
                if (g != 0.f) {
                    if (!MERGE) {
                        unsafeAtomicAdd(gvbase + c.row[k] * stride, g * go);
                    } else {
                        const int grow = brow + (int)c.row[k];
                        if (grow == prow[k]) {
                            pa[k] += g * go;
                        } else {
                            if (prow[k] >= 0) unsafeAtomicAdd(gvbase0 + (int64_t)prow[k] * stride, pa[k]);
                            prow[k] = grow; pa[k] = g * go;
                        }
                    }
                }
            }
            if (MERGE && (p + 1 == P || lv.start[p + 1] != lv.start[p])) {      // end of a level (uniform): merge over the wave, flush
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int r = prow[k];
                    float a = pa[k];
#pragma unroll
                    for (int o = LPT; o < 64; o <<= 1) {
                        const int r1 = __shfl_xor(r, o, 64);
                        const float a1 = __shfl_xor(a, o, 64);
                        const bool partner = (int)((threadIdx.x ^ o) / LPT) < nq;
                        if (partner && r >= 0 && r1 == r) {
                            if (threadIdx.x & o) r = -1;
                            else a += a1;
                        }
                    }
                    if (r >= 0) unsafeAtomicAdd(gvbase0 + (int64_t)r * stride, a);
                    prow[k] = -1;
                }
            }
            float gw = c.wy0 * c.wx0 * d[0] + c.wy0 * c.wx1 * d[1] + c.wy1 * c.wx0 * d[2] + c.wy1 * c.wx1 * d[3];
            float gx = (c.wy0 * (d[1] - d[0]) + c.wy1 * (d[3] - d[2])) * aw * (float)lv.w[p];
            float gy = (c.wx0 * (d[2] - d[0]) + c.wx1 * (d[3] - d[1])) * aw * (float)lv.h[p];
            gw = group_sum<LPT>(gw);
            gx = group_sum<LPT>(gx);
            gy = group_sum<LPT>(gy);
            dot_acc += aw * gw;
            if (ch == 0) {
                s_gx[qi * P + p] = gx; s_gy[qi * P + p] = gy; s_gw[qi * P + p] = gw;
                if (FUSED) pw[p] = aw;
            }
        }
        if (ch == 0) s_dot[qi] = dot_acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq * P; i += kThreads) {
        const int qj = i / P, p = i - qj * P;
        const int64_t task = (int64_t)(q0 + qj) * H + head;
        if (FUSED) {
            const float4 r = *reinterpret_cast<const float4 *>(ref + (int64_t)(q0 + qj) * 4);
            const float sc = lv.inv_n[p] * offset_scale;
            store_f(grad_offsets + (task * P + p) * 2, s_gx[i] * sc * r.z);
            store_f(grad_offsets + (task * P + p) * 2 + 1, s_gy[i] * sc * r.w);
            store_f(grad_logits + task * P + p, s_w[i] * (s_gw[i] - s_dot[qj]));
        } else {
            grad_loc[(task * P + p) * 2] = s_gx[i];
            grad_loc[(task * P + p) * 2 + 1] = s_gy[i];
            grad_weight[task * P + p] = s_gw[i];
        }
    }
}

// Backward, "pair" mapping: D / 2 lanes per task, TWO channels per lane, so that one 8-byte integer atomic carries both:
// d(value) is accumulated in int32 fixed point (contribution * fx_scale, rounded to nearest), channel pair (2j, 2j+1) packed
// as lo + hi * 2^32 in one int64 and added with global_atomic_add_x2.  Integer addition is exact and commutative - the
// result does not depend on the order the atomics retire in (f32 atomics do) - and the carry of the signed low half into
// the high half is undone exactly when the buffer is read back: lo = (int32)R, hi = (R - lo) >> 32.  The L2 atomic units
// retire one LANE operation per clock per channel (tools/msda_acc_bench.py: f32 599 us, int32 482 us, this kernel: see
// DESIGN.md), so two channels per lane-op is what halves the time; a (task, corner) still covers one 128-byte line.
// AM 2: the same mapping with ONE packed f16 atomic per channel pair (global_atomic_pk_add_f16) into an f16 accumulator whose
// power-of-two scale (fx_state) keeps the largest possible sum below 2^15: the atomic units retire one DWORD per clock per
// channel (f32 608 us, int32 482 us, int64 pairs 548 us, packed 16-bit pairs 325 us at the bench shape), so halving the
// dwords is what halves the time.  f16 carries 11 significant bits (bf16: 8, measured 1.9 % of max worst-case error and
// addends below 1/256 of a running sum lost outright) and, scaled, 39 binary orders below the largest sum.
// MERGE: contributions that meet in one value row are summed in registers before they reach the atomic units - over the
// consecutive points of a level inside a lane (a pending (row, sum) pair per corner, flushed when the row changes) and, at
// the end of every level, over the tasks of the wave (butterfly over lane ^ LPT, ^ 2 LPT, ...: the lower task takes the sum,
// the upper one drops out).  In training a fifth of the queries are the PADDING entries of the denoising groups: zero boxes
// whose 12 points all sit on pixel (0, 0) of their level, consecutive in the query order - 12 x 4 atomics of a wave on three
// hot rows become three.  Every contribution is still added exactly once; sums are formed in fp32 before the f16 / integer
// conversion.
template <typename T, int D, bool FUSED, int AM, bool MERGE>
__global__ __launch_bounds__(kThreads) void msda_bwd_pair_kernel(
    const T *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ weight,
    const float *__restrict__ ref, const T *__restrict__ offsets, const T *__restrict__ logits,
    const T *__restrict__ grad_out, void *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_weight, T *__restrict__ grad_offsets, T *__restrict__ grad_logits,
    MsdaLevels lv, int L, int H, int Lq, int total_q, float offset_scale, const float *__restrict__ fx_scale_p) {
    constexpr int LPT = D / 2;
    constexpr int QPB = kThreads / LPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = lv.n_points;
    float *s_x = smem, *s_y = smem + QPB * P, *s_w = smem + 2 * QPB * P;
    float *s_gx = smem + 3 * QPB * P, *s_gy = smem + 4 * QPB * P, *s_gw = smem + 5 * QPB * P;
    float *s_dot = smem + 6 * QPB * P;

    const int head = blockIdx.x % H;
    const int q0 = (blockIdx.x / H) * QPB;
    const int nq = min(QPB, total_q - q0);
    stage_points<T, FUSED>(lv, loc, weight, ref, offsets, logits, offset_scale, head, H, q0, nq,
                           total_q, s_x, s_y, s_w);
    const float fx_scale = *fx_scale_p;
    __syncthreads();

    const int qi = threadIdx.x / LPT, c2 = (threadIdx.x % LPT) * 2;
    if (qi < nq) {
        const int gq = q0 + qi;
        const int b = gq / Lq;
        const int64_t stride = (int64_t)H * D;
        const int64_t base = ((int64_t)b * L * H + head) * D + c2;
        const T *vbase = value + base;
        float *px = s_x + qi * P, *py = s_y + qi * P, *pw = s_w + qi * P;
        float go0, go1;
        if (sizeof(T) == 2) {
            const uint32_t gg = *reinterpret_cast<const uint32_t *>(grad_out + ((int64_t)gq * H + head) * D + c2);
            go0 = __uint_as_float(gg << 16); go1 = __uint_as_float(gg & 0xffff0000u);
        } else {
            const float2 gg = *reinterpret_cast<const float2 *>(grad_out + ((int64_t)gq * H + head) * D + c2);
            go0 = gg.x; go1 = gg.y;
        }
        const float gs0 = go0 * fx_scale, gs1 = go1 * fx_scale;

        float wmax = 0.f, winv = 1.f;
        if (FUSED) {
            wmax = pw[0];
            for (int p = 1; p < P; ++p) wmax = fmaxf(wmax, pw[p]);
            float s = 0.f;
            for (int p = 0; p < P; ++p) s += __expf(pw[p] - wmax);
            winv = 1.f / s;
        }
        // element offset of (row 0, this head, this channel pair) of image 0; a pending row is kept as b * L + row
        const int64_t base0 = (int64_t)head * D + c2;
        const int brow = b * L;
        auto add_pair = [&](int grow, float a0, float a1) {
            const int64_t e = base0 + (int64_t)grow * stride;
            if (AM == 3) {
                const long long lo = (long long)__float2int_rn(a0), hi = (long long)__float2int_rn(a1);
                long long *dst = reinterpret_cast<long long *>(reinterpret_cast<int *>(grad_value) + e);
                __hip_atomic_fetch_add(dst, lo + (hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const half2_t v = {(_Float16)a0, (_Float16)a1};
                _Float16 *dst = reinterpret_cast<_Float16 *>(grad_value) + e;
                __builtin_amdgcn_global_atomic_fadd_v2f16((half2_t __attribute__((address_space(1))) *)dst, v);
            }
        };
        int prow[4] = {-1, -1, -1, -1};
        float pa0[4] = {0.f, 0.f, 0.f, 0.f}, pa1[4] = {0.f, 0.f, 0.f, 0.f};
        float dot_acc = 0.f;
        for (int p = 0; p < P; ++p) {
            const Corner c = corners(px[p], py[p], lv.start[p], lv.h[p], lv.w[p]);
            const float aw = FUSED ? __expf(pw[p] - wmax) * winv : pw[p];
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v0, v1;
                if (sizeof(T) == 2) {
                    const uint32_t vv = *reinterpret_cast<const uint32_t *>(vbase + c.row[k] * stride);
                    v0 = __uint_as_float(vv << 16); v1 = __uint_as_float(vv & 0xffff0000u);
                } else {
                    const float2 vv = *reinterpret_cast<const float2 *>(vbase + c.row[k] * stride);
                    v0 = vv.x; v1 = vv.y;
                }
                d[k] = (go0 * v0 + go1 * v1) * c.ok[k];
                const float g = aw * c.bw[k];
                if (g != 0.f) {
                    const int grow = brow + (int)c.row[k];
                    if (!MERGE) {
                        add_pair(grow, g * gs0, g * gs1);
                    } else if (grow == prow[k]) {
                        pa0[k] += g * gs0; pa1[k] += g * gs1;
                    } else {
                        if (prow[k] >= 0) add_pair(prow[k], pa0[k], pa1[k]);
                        prow[k] = grow; pa0[k] = g * gs0; pa1[k] = g * gs1;
                    }
                }
            }
            if (MERGE && (p + 1 == P || lv.start[p + 1] != lv.start[p])) {      // end of a level (uniform): merge over the wave, flush
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int r = prow[k];
                    float a0 = pa0[k], a1 = pa1[k];
#pragma unroll
                    for (int o = LPT; o < 64; o <<= 1) {
                        const int r1 = __shfl_xor(r, o, 64);
                        const float b0 = __shfl_xor(a0, o, 64), b1 = __shfl_xor(a1, o, 64);
                        const bool partner = (int)((threadIdx.x ^ o) / LPT) < nq;    // lanes of tasks past the last query hold nothing
                        if (partner && r >= 0 && r1 == r) {
                            if (threadIdx.x & o) r = -1;
                            else { a0 += b0; a1 += b1; }
                        }
                    }
                    if (r >= 0) add_pair(r, a0, a1);
                    prow[k] = -1;
                }
            }
            float gw = c.wy0 * c.wx0 * d[0] + c.wy0 * c.wx1 * d[1] + c.wy1 * c.wx0 * d[2] + c.wy1 * c.wx1 * d[3];
            float gx = (c.wy0 * (d[1] - d[0]) + c.wy1 * (d[3] - d[2])) * aw * (float)lv.w[p];
            float gy = (c.wx0 * (d[2] - d[0]) + c.wx1 * (d[3] - d[1])) * aw * (float)lv.h[p];
            gw = group_sum<LPT>(gw);
            gx = group_sum<LPT>(gx);
            gy = group_sum<LPT>(gy);
            dot_acc += aw * gw;
            if (c2 == 0) {
                s_gx[qi * P + p] = gx; s_gy[qi * P + p] = gy; s_gw[qi * P + p] = gw;
                if (FUSED) pw[p] = aw;
            }
        }
        if (c2 == 0) s_dot[qi] = dot_acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq * P; i += kThreads) {
        const int qj = i / P, p = i - qj * P;
        const int64_t task = (int64_t)(q0 + qj) * H + head;
        if (FUSED) {
            const float4 r = *reinterpret_cast<const float4 *>(ref + (int64_t)(q0 + qj) * 4);
            const float sc = lv.inv_n[p] * offset_scale;
            store_f(grad_offsets + (task * P + p) * 2, s_gx[i] * sc * r.z);
            store_f(grad_offsets + (task * P + p) * 2 + 1, s_gy[i] * sc * r.w);
            store_f(grad_logits + task * P + p, s_w[i] * (s_gw[i] - s_dot[qj]));
        } else {
            grad_loc[(task * P + p) * 2] = s_gx[i];
            grad_loc[(task * P + p) * 2 + 1] = s_gy[i];
            grad_weight[task * P + p] = s_gw[i];
        }
    }
}

__global__ void cast_f32_bf16_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t step = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += step) {
        const float4 v = *reinterpret_cast<const float4 *>(src + i);
        Vec4<uint16_t>::store(dst + i, {v.x, v.y, v.z, v.w});
    }
    if (i < n && i + 3 >= n) for (int64_t j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
}

// ---- scale state of the fixed-point d(value) accumulator ------------------------------------------------------------
// fx_state (device, 4 words, zero-filled together with the accumulator): [0] f32 scale (fixed-point units per 1.0),
// [1] f32 capacity = the largest |grad_out| the scale was sized for, [2] u32 bits of max |grad_out| of the current call,
// [3] i32 right-shift the accumulator needs before the current call adds into it.
// Bound: every (query, head) spreads softmax weights (sum 1) x bilinear weights (<= 1) over its corners, so one
// accumulator word receives at most |grad_out| per query and call: |sum| <= hit_bound * capacity with hit_bound =
// (calls sharing the accumulator) x Lq, kept below 2^30.  A later call with larger gradients than the first one (x 4 head
// room) rescales the accumulator in place (rare: one extra pass over it).
template <typename T>
__global__ __launch_bounds__(256) void fx_gmax_kernel(const T *__restrict__ g, int64_t n, float *__restrict__ state) {
    float m = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = Vec4<T>::load(g + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(load_f(g + (n & ~(int64_t)3) + threadIdx.x)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m == m) atomicMax(reinterpret_cast<unsigned *>(state) + 2, __float_as_uint(fminf(m, 1e30f)));   // non-negative floats order like their bits
    }
}

__global__ void fx_update_kernel(float *__restrict__ state, float hit_bound, float limit_log2) {
    unsigned *su = reinterpret_cast<unsigned *>(state);
    int *si = reinterpret_cast<int *>(state);
    float gmax = __uint_as_float(su[2]);
    su[2] = 0u;
    gmax = fmaxf(gmax, 1e-30f);
    int shift = 0;
    if (state[0] == 0.f) {
        const float cap = exp2f(ceilf(log2f(gmax)) + 2.f);           // power of two >= 4 gmax: scales stay exact powers of two
        state[1] = cap;
        state[0] = exp2f(limit_log2) / (exp2f(ceilf(log2f(fmaxf(hit_bound, 1.f)))) * cap);
    } else if (gmax > state[1]) {
        shift = (int)ceilf(log2f(gmax / state[1])) + 1;
        if (shift > 62) shift = 62;
        state[1] *= exp2f((float)shift);
        state[0] *= exp2f(-(float)shift);
    }
    si[3] = shift;
}

// acc holds channel pairs lo + hi * 2^32 (see msda_bwd_pair_kernel): shift both halves right with rounding
__global__ __launch_bounds__(256) void fx_rescale_kernel(long long *__restrict__ acc, int64_t n, const float *__restrict__ state) {
    const int shift = reinterpret_cast<const int *>(state)[3];
    if (shift == 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long r = acc[i];
        const long long lo = (long long)(int)(r & 0xffffffffll), hi = (r - lo) >> 32;
        const long long half = 1ll << (shift - 1);
        const long long lo2 = (lo + half) >> shift, hi2 = (hi + half) >> shift;
        acc[i] = lo2 + hi2 * 4294967296ll;
    }
}

__global__ __launch_bounds__(256) void fx_rescale_f16_kernel(half2_t *__restrict__ acc, int64_t n, const float *__restrict__ state) {
    const int shift = reinterpret_cast<const int *>(state)[3];
    if (shift == 0) return;
    const _Float16 f = (_Float16)exp2f(-(float)min(shift, 24));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        half2_t v = acc[i];
        v.x *= f; v.y *= f;
        acc[i] = v;
    }
}

template <typename T>
__global__ void cast_f16acc_kernel(const _Float16 *__restrict__ src, T *__restrict__ dst, int64_t n, const float *__restrict__ state) {
    const float inv = 1.f / state[0];
    typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i + 3 < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const half4_t v = *reinterpret_cast<const half4_t *>(src + i);
        Vec4<T>::store(dst + i, {(float)v.x * inv, (float)v.y * inv, (float)v.z * inv, (float)v.w * inv});
    }
    if (i < n && i + 3 >= n) for (int64_t j = i; j < n; ++j) store_f(dst + j, (float)src[j] * inv);
}

template <typename T>
__global__ void cast_fixed_kernel(const long long *__restrict__ src, T *__restrict__ dst, int64_t npairs, const float *__restrict__ state) {
    const float inv = 1.f / state[0];
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < npairs; i += (int64_t)gridDim.x * blockDim.x * 2) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long long r = i + k < npairs ? src[i + k] : 0ll;
            const long long lo = (long long)(int)(r & 0xffffffffll), hi = (r - lo) >> 32;
            o[2 * k] = (float)lo * inv; o[2 * k + 1] = (float)hi * inv;
        }
        if (i + 1 < npairs) Vec4<T>::store(dst + 2 * i, {o[0], o[1], o[2], o[3]});
        else { store_f(dst + 2 * i, o[0]); store_f(dst + 2 * i + 1, o[1]); }
    }
}

// -------------------------------------------------------------------------------------------
static int fill_levels(MsdaLevels &lv, int L, int n_levels, const int *hw, const int *pts) {
    if (n_levels < 1 || n_levels > DFINE_MAX_LEVELS || !hw || !pts) return DFINE_E_BADARG;
    int p = 0, row = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int h = hw[2 * l], w = hw[2 * l + 1], n = pts[l];
        if (h < 1 || w < 1 || n < 1 || p + n > DFINE_MAX_POINTS) return DFINE_E_BADARG;
        for (int k = 0; k < n; ++k, ++p) {
            lv.start[p] = row; lv.h[p] = h; lv.w[p] = w; lv.inv_n[p] = 1.0f / (float)n;
        }
        row += h * w;
    }
    if (row != L) return DFINE_E_BADARG;
    lv.n_points = p;
    return DFINE_OK;
}

template <typename T, bool FUSED>
static int launch_fwd(const void *value, const float *loc, const float *weight, const float *ref,
                      const void *offsets, const void *logits, void *out, int B, int L, int H,
                      int D, int Lq, const MsdaLevels &lv, float offset_scale, hipStream_t st) {
    const int total_q = B * Lq;
    if (total_q == 0) return DFINE_OK;
    if (sizeof(T) == 2 && (D == 32 || D == 16 || D == 64)) {
#define DFINE_FWD8(DD)                                                                         \
    {                                                                                          \
        constexpr int QPB = kThreads / (DD / 8);                                               \
        const int nblk = ((total_q + QPB - 1) / QPB) * H;                                      \
        const size_t sm = sizeof(float) * 3 * QPB * lv.n_points;                               \
        hipLaunchKernelGGL((msda_fwd8_kernel<DD, FUSED>), dim3(nblk), dim3(kThreads), sm, st,  \
                           (const uint16_t *)value, loc, weight, ref, (const uint16_t *)offsets, \
                           (const uint16_t *)logits, (uint16_t *)out, lv, L, H, Lq, total_q, offset_scale); \
    }
        if (D == 32) DFINE_FWD8(32) else if (D == 16) DFINE_FWD8(16) else DFINE_FWD8(64)
#undef DFINE_FWD8
        return check_launch();
    }
#define DFINE_FWD(DD)                                                                          \
    {                                                                                          \
        constexpr int QPB = kThreads / (DD / 4);                                               \
        const int nblk = ((total_q + QPB - 1) / QPB) * H;                                      \
        const size_t sm = sizeof(float) * 3 * QPB * lv.n_points;                               \
        hipLaunchKernelGGL((msda_fwd_kernel<T, DD, FUSED>), dim3(nblk), dim3(kThreads), sm, st, \
                           (const T *)value, loc, weight, ref, (const T *)offsets,             \
                           (const T *)logits, (T *)out, lv, L, H, Lq, total_q, offset_scale);  \
    }
    if (D == 32) DFINE_FWD(32) else if (D == 16) DFINE_FWD(16) else if (D == 64) DFINE_FWD(64)
    else return DFINE_E_BADARG;
#undef DFINE_FWD
    return check_launch();
}

template <typename T, bool FUSED>
static int launch_bwd(const void *value, const float *loc, const float *weight, const float *ref,
                      const void *offsets, const void *logits, const void *grad_out,
                      float *grad_value, float *grad_loc, float *grad_weight, void *grad_offsets,
                      void *grad_logits, int B, int L, int H, int D, int Lq, const MsdaLevels &lv,
                      float offset_scale, hipStream_t st, int acc_mode = 0, float *fx_state = nullptr, float hit_bound = 1.f) {
    const int total_q = B * Lq;
    if (total_q == 0) return DFINE_OK;
    // (merge: coinciding rows are merged in registers before the atomics, see the kernels)
    const bool merge = (int64_t)B * L < (int64_t)1 << 31;
    if (acc_mode == 2 || acc_mode == 3) {
        if (!(D == 32 || D == 16 || D == 64)) return DFINE_E_BADARG;
        // scale bookkeeping of the scaled accumulator (see fx_update_kernel): three small launches in front of the gather
        hipLaunchKernelGGL((fx_gmax_kernel<T>), dim3(256), dim3(256), 0, st, (const T *)grad_out, (int64_t)total_q * H * D, fx_state);
        hipLaunchKernelGGL(fx_update_kernel, dim3(1), dim3(1), 0, st, fx_state, hit_bound, acc_mode == 3 ? 30.f : 15.f);
        if (acc_mode == 3)
            hipLaunchKernelGGL(fx_rescale_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<long long *>(grad_value),
                               (int64_t)B * L * H * D / 2, fx_state);
        else
            hipLaunchKernelGGL(fx_rescale_f16_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<half2_t *>(grad_value),
                               (int64_t)B * L * H * D / 2, fx_state);
#define DFINE_BWDP(DD, AM)                                                                     \
    {                                                                                          \
        constexpr int QPB = kThreads / (DD / 2);                                               \
        const int nblk = ((total_q + QPB - 1) / QPB) * H;                                      \
        const size_t sm = sizeof(float) * (6 * QPB * lv.n_points + QPB);                       \
        if (merge)                                                                             \
            hipLaunchKernelGGL((msda_bwd_pair_kernel<T, DD, FUSED, AM, true>), dim3(nblk), dim3(kThreads), sm, st, \
                               (const T *)value, loc, weight, ref, (const T *)offsets,         \
                               (const T *)logits, (const T *)grad_out, (void *)grad_value, grad_loc, \
                               grad_weight, (T *)grad_offsets, (T *)grad_logits, lv, L, H, Lq, \
                               total_q, offset_scale, (const float *)fx_state);                \
        else                                                                                   \
            hipLaunchKernelGGL((msda_bwd_pair_kernel<T, DD, FUSED, AM, false>), dim3(nblk), dim3(kThreads), sm, st, \
                               (const T *)value, loc, weight, ref, (const T *)offsets,         \
                               (const T *)logits, (const T *)grad_out, (void *)grad_value, grad_loc, \
                               grad_weight, (T *)grad_offsets, (T *)grad_logits, lv, L, H, Lq, \
                               total_q, offset_scale, (const float *)fx_state);                \
    }
        if (acc_mode == 3) { if (D == 32) DFINE_BWDP(32, 3) else if (D == 16) DFINE_BWDP(16, 3) else DFINE_BWDP(64, 3) }
        else { if (D == 32) DFINE_BWDP(32, 2) else if (D == 16) DFINE_BWDP(16, 2) else DFINE_BWDP(64, 2) }
#undef DFINE_BWDP
        return check_launch();
    }
    if (acc_mode != 0) return DFINE_E_BADARG;
    if (D == 32 || D == 16 || D == 64) {
#define DFINE_BWDW(DD)                                                                         \
    {                                                                                          \
        constexpr int QPB = kThreads / DD;                                                     \
        const int nblk = ((total_q + QPB - 1) / QPB) * H;                                      \
        const size_t sm = sizeof(float) * (6 * QPB * lv.n_points + QPB);                       \
        if (merge)                                                                             \
            hipLaunchKernelGGL((msda_bwd_wide_kernel<T, DD, FUSED, true>), dim3(nblk), dim3(kThreads), sm, st, \
                               (const T *)value, loc, weight, ref, (const T *)offsets,         \
                               (const T *)logits, (const T *)grad_out, grad_value, grad_loc,   \
                               grad_weight, (T *)grad_offsets, (T *)grad_logits, lv, L, H, Lq, \
                               total_q, offset_scale);                                         \
        else                                                                                   \
            hipLaunchKernelGGL((msda_bwd_wide_kernel<T, DD, FUSED, false>), dim3(nblk), dim3(kThreads), sm, st, \
                               (const T *)value, loc, weight, ref, (const T *)offsets,         \
                               (const T *)logits, (const T *)grad_out, grad_value, grad_loc,   \
                               grad_weight, (T *)grad_offsets, (T *)grad_logits, lv, L, H, Lq, \
                               total_q, offset_scale);                                         \
    }
        if (D == 32) DFINE_BWDW(32) else if (D == 16) DFINE_BWDW(16) else DFINE_BWDW(64)
#undef DFINE_BWDW
        return check_launch();
    }
#define DFINE_BWD(DD)                                                                          \
    {                                                                                          \
        constexpr int QPB = kThreads / (DD / 4);                                               \
        const int nblk = ((total_q + QPB - 1) / QPB) * H;                                      \
        const size_t sm = sizeof(float) * (6 * QPB * lv.n_points + QPB);                       \
        hipLaunchKernelGGL((msda_bwd_kernel<T, DD, FUSED>), dim3(nblk), dim3(kThreads), sm, st, \
                           (const T *)value, loc, weight, ref, (const T *)offsets,             \
                           (const T *)logits, (const T *)grad_out, grad_value, grad_loc,       \
                           grad_weight, (T *)grad_offsets, (T *)grad_logits, lv, L, H, Lq,     \
                           total_q, offset_scale);                                             \
    }
    if (D == 32) DFINE_BWD(32) else if (D == 16) DFINE_BWD(16) else if (D == 64) DFINE_BWD(64)
    else return DFINE_E_BADARG;
#undef DFINE_BWD
    return check_launch();
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_msda_fwd(const void *value, const float *loc, const float *weight, void *out, int dtype,
                   int B, int L, int H, int D, int Lq, int n_levels, const int *level_hw,
                   const int *level_points, void *stream) {
    if (B == 0 || Lq == 0) return DFINE_OK;   // empty problem: nothing to launch
    if (!value || !loc || !weight || !out || B < 0 || Lq < 0 || H < 1) return DFINE_E_BADARG;
    MsdaLevels lv;
    if (int e = fill_levels(lv, L, n_levels, level_hw, level_points)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        return launch_fwd<float, false>(value, loc, weight, nullptr, nullptr, nullptr, out, B, L, H, D, Lq, lv, 0.f, st);
    if (dtype == DFINE_BF16)
        return launch_fwd<uint16_t, false>(value, loc, weight, nullptr, nullptr, nullptr, out, B, L, H, D, Lq, lv, 0.f, st);
    return DFINE_E_BADARG;
}

int dfine_msda_bwd(const void *value, const float *loc, const float *weight, const void *grad_out,
                   float *grad_value_f32, float *grad_loc, float *grad_weight, int dtype, int B,
                   int L, int H, int D, int Lq, int n_levels, const int *level_hw,
                   const int *level_points, void *stream) {
    if (B == 0 || Lq == 0) return DFINE_OK;   // empty problem: nothing to launch
    if (!value || !loc || !weight || !grad_out || !grad_value_f32 || !grad_loc || !grad_weight ||
        B < 0 || Lq < 0 || H < 1)
        return DFINE_E_BADARG;
    MsdaLevels lv;
    if (int e = fill_levels(lv, L, n_levels, level_hw, level_points)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        return launch_bwd<float, false>(value, loc, weight, nullptr, nullptr, nullptr, grad_out, grad_value_f32,
                                        grad_loc, grad_weight, nullptr, nullptr, B, L, H, D, Lq, lv, 0.f, st);
    if (dtype == DFINE_BF16)
        return launch_bwd<uint16_t, false>(value, loc, weight, nullptr, nullptr, nullptr, grad_out, grad_value_f32,
                                           grad_loc, grad_weight, nullptr, nullptr, B, L, H, D, Lq, lv, 0.f, st);
    return DFINE_E_BADARG;
}

int dfine_msda_fused_fwd(const void *value, const float *ref, const void *offsets, const void *logits,
                         void *out, int dtype, int B, int L, int H, int D, int Lq, int n_levels,
                         const int *level_hw, const int *level_points, float offset_scale, void *stream) {
    if (B == 0 || Lq == 0) return DFINE_OK;   // empty problem: nothing to launch
    if (!value || !ref || !offsets || !logits || !out || B < 0 || Lq < 0 || H < 1) return DFINE_E_BADARG;
    MsdaLevels lv;
    if (int e = fill_levels(lv, L, n_levels, level_hw, level_points)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        return launch_fwd<float, true>(value, nullptr, nullptr, ref, offsets, logits, out, B, L, H, D, Lq, lv, offset_scale, st);
    if (dtype == DFINE_BF16)
        return launch_fwd<uint16_t, true>(value, nullptr, nullptr, ref, offsets, logits, out, B, L, H, D, Lq, lv, offset_scale, st);
    return DFINE_E_BADARG;
}

int dfine_msda_fused_bwd(const void *value, const float *ref, const void *offsets, const void *logits,
                         const void *grad_out, float *grad_value_f32, void *grad_offsets,
                         void *grad_logits, int dtype, int B, int L, int H, int D, int Lq, int n_levels,
                         const int *level_hw, const int *level_points, float offset_scale, void *stream) {
    if (B == 0 || Lq == 0) return DFINE_OK;   // empty problem: nothing to launch
    if (!value || !ref || !offsets || !logits || !grad_out || !grad_value_f32 || !grad_offsets ||
        !grad_logits || B < 0 || Lq < 0 || H < 1)
        return DFINE_E_BADARG;
    MsdaLevels lv;
    if (int e = fill_levels(lv, L, n_levels, level_hw, level_points)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFINE_F32)
        return launch_bwd<float, true>(value, nullptr, nullptr, ref, offsets, logits, grad_out, grad_value_f32,
                                       nullptr, nullptr, grad_offsets, grad_logits, B, L, H, D, Lq, lv, offset_scale, st);
    if (dtype == DFINE_BF16)
        return launch_bwd<uint16_t, true>(value, nullptr, nullptr, ref, offsets, logits, grad_out, grad_value_f32,
                                          nullptr, nullptr, grad_offsets, grad_logits, B, L, H, D, Lq, lv, offset_scale, st);
    return DFINE_E_BADARG;
}

// acc_mode: 0 = grad_value_acc is f32 (hardware f32 atomics, as dfine_msda_fused_bwd); 2 = f16, one packed f16 atomic per
// channel pair, power-of-two scale in fx_state; 3 = int32 fixed point, channel pairs packed in int64, integer atomics (exact,
// order-independent).  Modes 2 and 3 are finished by dfine_cast_scaled_acc.
int dfine_msda_fused_bwd_acc(const void *value, const float *ref, const void *offsets, const void *logits,
                             const void *grad_out, void *grad_value_acc, void *grad_offsets,
                             void *grad_logits, int dtype, int B, int L, int H, int D, int Lq, int n_levels,
                             const int *level_hw, const int *level_points, float offset_scale, int acc_mode,
                             float *fx_state, float hit_bound, void *stream) {
    if (B == 0 || Lq == 0) return DFINE_OK;
    if (!value || !ref || !offsets || !logits || !grad_out || !grad_value_acc || !grad_offsets ||
        !grad_logits || B < 0 || Lq < 0 || H < 1 || (acc_mode != 0 && acc_mode != 2 && acc_mode != 3) ||
        (acc_mode != 0 && (!fx_state || !(hit_bound >= 1.f))))
        return DFINE_E_BADARG;
    MsdaLevels lv;
    if (int e = fill_levels(lv, L, n_levels, level_hw, level_points)) return e;
    hipStream_t st = (hipStream_t)stream;
    float *acc = reinterpret_cast<float *>(grad_value_acc);
    if (dtype == DFINE_F32)
        return launch_bwd<float, true>(value, nullptr, nullptr, ref, offsets, logits, grad_out, acc,
                                       nullptr, nullptr, grad_offsets, grad_logits, B, L, H, D, Lq, lv, offset_scale, st,
                                       acc_mode, fx_state, hit_bound);
    if (dtype == DFINE_BF16)
        return launch_bwd<uint16_t, true>(value, nullptr, nullptr, ref, offsets, logits, grad_out, acc,
                                          nullptr, nullptr, grad_offsets, grad_logits, B, L, H, D, Lq, lv, offset_scale, st,
                                          acc_mode, fx_state, hit_bound);
    return DFINE_E_BADARG;
}

int dfine_cast_scaled_acc(const void *src, void *dst, int acc_mode, int dtype, int64_t n, const float *fx_state, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!src || !dst || !fx_state || n < 0 || (n & 1) || (dtype != DFINE_BF16 && dtype != DFINE_F32) || (acc_mode != 2 && acc_mode != 3))
        return DFINE_E_BADARG;
    const int threads = 256;
    int64_t blocks = (n / 4 + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (acc_mode == 3) {
        if (dtype == DFINE_BF16)
            hipLaunchKernelGGL(cast_fixed_kernel<uint16_t>, dim3((unsigned)blocks), dim3(threads), 0, st, (const long long *)src, (uint16_t *)dst, n / 2, fx_state);
        else
            hipLaunchKernelGGL(cast_fixed_kernel<float>, dim3((unsigned)blocks), dim3(threads), 0, st, (const long long *)src, (float *)dst, n / 2, fx_state);
    } else {
        if (dtype == DFINE_BF16)
            hipLaunchKernelGGL(cast_f16acc_kernel<uint16_t>, dim3((unsigned)blocks), dim3(threads), 0, st, (const _Float16 *)src, (uint16_t *)dst, n, fx_state);
        else
            hipLaunchKernelGGL(cast_f16acc_kernel<float>, dim3((unsigned)blocks), dim3(threads), 0, st, (const _Float16 *)src, (float *)dst, n, fx_state);
    }
    return check_launch();
}

int dfine_cast_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!src || !dst || n < 0) return DFINE_E_BADARG;
    const int threads = 256;
    int64_t blocks = (n / 4 + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                       (hipStream_t)stream, src, (uint16_t *)dst, n);
    return check_launch();
}

}  // extern "C"
