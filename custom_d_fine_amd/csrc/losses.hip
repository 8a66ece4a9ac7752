// A13/A14 - fused set-criterion losses of one prediction head, forward values AND gradients in
// one pass (the gradient of every term is closed-form, so the backward of the autograd node is a
// scale of the stored buffers).
//
// Reference: DFINECriterion.loss_labels_vfl / loss_boxes / loss_local / unimodal_distribution_
// focal_loss (src/d_fine/dfine_criterion.py:92-237,837-858) with bbox2distance / translate_gt /
// weighting_function (src/d_fine/arch/utils.py:145-188,267-354) and box_iou /
// generalized_box_iou (src/d_fine/arch/utils.py:12-51).  The reference issues ~20 small ATen
// kernels per loss term, 48 terms per step for D-FINE-m - the criterion is host-bound (28 ms of
// launch overhead per step on MI355X).  Here one head costs one fill and two launches:
//   pair_box_kernel     per matched (image, query, target) of both matchings: IoU, GIoU, L1 (+ gradients wrt the predicted
//                       box) and the (image, query) -> pair maps
//   head_phase2_kernel  block roles: varifocal loss over all B*Q*C logits | T^2 * KL(softmax(teacher/T) || softmax(pred/T))
//                       over all B*Q*4 edge rows, weighted per (image, query) by the pair's IoU or sigmoid(max_c teacher_logit)
//                       [DDF] | fine-grained localisation CE on the matched rows, targets derived in-kernel [FGL]
// Inputs may be strided views (batch / query strides) of fp32 or bf16 tensors; math is fp32.
// Loss sums are accumulated with one atomic per block into out[] (zeroed by the entry point).
#include <cstdlib>

#include "common.h"

namespace dfine {

constexpr int kLT = 256;

__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < kLT / 64; ++w) s += red[w];
    __syncthreads();
    return s;   // valid on thread 0
}

struct View {            // [B, Q, inner] tensor view with unit inner stride
    int64_t sb, sq;      // element strides of the batch and query dims
};

// ---------------------------------------------------------------------------------------------
// plan: int64 [3, M] = (image, query, target row).  boxes cxcywh.
// One launch for both matchings of a head: blocks [0, nb_cls) walk the classification plan (IoU + map only), the others the
// box plan (IoU, map, L1 / GIoU sums and gradients).  map[b * Q + q] = pair index + 1; 0 = unmatched (zero-filled).
template <typename T>
__global__ __launch_bounds__(kLT) void pair_box_kernel(
    const T *__restrict__ boxes, View bv, const float *__restrict__ tgt_boxes,
    const int64_t *__restrict__ cls_plan, int M_cls, float *__restrict__ iou_cls, int *__restrict__ map_cls, int nb_cls,
    const int64_t *__restrict__ box_plan, int M_box, float *__restrict__ iou_box, int *__restrict__ map_box, int Q,
    float *__restrict__ grad_l1, float *__restrict__ grad_giou,
    float *__restrict__ out /* [0]+=l1 sum, [1]+=(1-giou) sum */, float s_l1, float s_giou,
    const float *__restrict__ dev_scales, const int *__restrict__ dev_m_box) {
    __shared__ float red[kLT / 64];
    if (dev_scales) { s_l1 = dev_scales[1]; s_giou = dev_scales[2]; }   // normalisers computed on the device (csrc/plans.hip)
    const bool is_cls = (int)blockIdx.x < nb_cls;                       // uniform per block
    const int64_t *__restrict__ plan = is_cls ? cls_plan : box_plan;
    const int M = is_cls ? M_cls : M_box;                               // row stride of the plan
    const int count = (!is_cls && dev_m_box) ? *dev_m_box : M;          // valid entries (the GO plan's length lives on the device)
    float *__restrict__ iou_out = is_cls ? iou_cls : iou_box;
    int *__restrict__ map = is_cls ? map_cls : map_box;
    const int with_loss = is_cls ? 0 : 1;
    const int m = ((int)blockIdx.x - (is_cls ? 0 : nb_cls)) * kLT + threadIdx.x;
    float l1 = 0.f, lg = 0.f;
    if (m < count) {
        const int64_t b = plan[m], q = plan[M + m], t = plan[2 * (int64_t)M + m];
        const T *sp = boxes + b * bv.sb + q * bv.sq;
        const float cx = load_f(sp), cy = load_f(sp + 1), w = load_f(sp + 2), h = load_f(sp + 3);
        const float4 tb = *reinterpret_cast<const float4 *>(tgt_boxes + t * 4);
        const float hw = 0.5f * fmaxf(w, 0.f), hh = 0.5f * fmaxf(h, 0.f);
        const float ax0 = cx - hw, ay0 = cy - hh, ax1 = cx + hw, ay1 = cy + hh;
        const float tw = 0.5f * fmaxf(tb.z, 0.f), th = 0.5f * fmaxf(tb.w, 0.f);
        const float bx0 = tb.x - tw, by0 = tb.y - th, bx1 = tb.x + tw, by1 = tb.y + th;
        const float aw = ax1 - ax0, ah = ay1 - ay0;
        const float area_a = aw * ah, area_b = (bx1 - bx0) * (by1 - by0);
        const float iwr = fminf(ax1, bx1) - fmaxf(ax0, bx0), ihr = fminf(ay1, by1) - fmaxf(ay0, by0);
        const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
        const float inter = iw * ih;
        const float uni = area_a + area_b - inter;
        const float iou = inter / uni;
        const float cwr = fmaxf(ax1, bx1) - fminf(ax0, bx0), chr_ = fmaxf(ay1, by1) - fminf(ay0, by0);
        const float cw = fmaxf(cwr, 0.f), ch = fmaxf(chr_, 0.f);
        const float hull = cw * ch;
        const float giou = iou - (hull - uni) / hull;
        if (iou_out) iou_out[m] = iou;
        if (map) map[b * Q + q] = m + 1;
        if (with_loss) {
            const float d0 = cx - tb.x, d1 = cy - tb.y, d2 = w - tb.z, d3 = h - tb.w;
            l1 = fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
            lg = 1.f - giou;
            float *g1 = grad_l1 + (b * Q + q) * 4;
            g1[0] = d0 > 0.f ? s_l1 : (d0 < 0.f ? -s_l1 : 0.f);
            g1[1] = d1 > 0.f ? s_l1 : (d1 < 0.f ? -s_l1 : 0.f);
            g1[2] = d2 > 0.f ? s_l1 : (d2 < 0.f ? -s_l1 : 0.f);
            g1[3] = d3 > 0.f ? s_l1 : (d3 < 0.f ? -s_l1 : 0.f);
            // d giou / d (ax0, ay0, ax1, ay1)
            const float ix = iwr > 0.f ? 1.f : 0.f, iy = ihr > 0.f ? 1.f : 0.f;
            const float di[4] = {ih * ix * (ax0 > bx0 ? -1.f : 0.f), iw * iy * (ay0 > by0 ? -1.f : 0.f),
                                 ih * ix * (ax1 < bx1 ? 1.f : 0.f), iw * iy * (ay1 < by1 ? 1.f : 0.f)};
            const float da[4] = {-ah, -aw, ah, aw};
            const float hx = cwr > 0.f ? 1.f : 0.f, hy = chr_ > 0.f ? 1.f : 0.f;
            const float dh[4] = {ch * hx * (ax0 < bx0 ? -1.f : 0.f), cw * hy * (ay0 < by0 ? -1.f : 0.f),
                                 ch * hx * (ax1 > bx1 ? 1.f : 0.f), cw * hy * (ay1 > by1 ? 1.f : 0.f)};
            float dg[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float du = da[k] - di[k];
                const float diou = (di[k] * uni - inter * du) / (uni * uni);
                dg[k] = diou + (du * hull - uni * dh[k]) / (hull * hull);
            }
            const float sw = w > 0.f ? 0.5f : 0.f, sh = h > 0.f ? 0.5f : 0.f;
            float *g2 = grad_giou + (b * Q + q) * 4;     // d (1 - giou) / d (cx, cy, w, h)
            g2[0] = -s_giou * (dg[0] + dg[2]);
            g2[1] = -s_giou * (dg[1] + dg[3]);
            g2[2] = -s_giou * (sw * (dg[2] - dg[0]));
            g2[3] = -s_giou * (sh * (dg[3] - dg[1]));
        }
    }
    if (with_loss) {
        const float s1 = block_sum(l1, red);
        const float s2 = block_sum(lg, red);
        if (threadIdx.x == 0) { unsafeAtomicAdd(out, s_l1 * s1); unsafeAtomicAdd(out + 1, s_giou * s2); }
    }
}

// ---------------------------------------------------------------------------------------------
// (the three loss bodies below are the block roles of head_phase2_kernel: `blk` of `nblk` blocks of kLT threads)
template <typename T>
__device__ __forceinline__ void vfl_body(int blk, int nblk, float *red, const T *__restrict__ logits, View lv,
                                         const int *__restrict__ map,
                                         const int64_t *__restrict__ plan, int M,
                                         const int64_t *__restrict__ labels,
                                         const float *__restrict__ iou, int B, int Q, int C,
                                         float alpha, float gamma, float s_vfl,
                                         T *__restrict__ grad, float *__restrict__ out) {
    const uint32_t n = (uint32_t)B * Q * C;               // < 2^31 (checked by the entry point): 32-bit index arithmetic
    float acc = 0.f;
    for (uint32_t e = (uint32_t)blk * kLT + threadIdx.x; e < n; e += (uint32_t)nblk * kLT) {
        const uint32_t row = e / (uint32_t)C;
        const int c = (int)(e - row * C);
        const uint32_t b = row / (uint32_t)Q, q = row - b * Q;
        const float x = load_f(logits + (int64_t)b * lv.sb + (int64_t)q * lv.sq + c);
        const float p = __builtin_amdgcn_rcpf(1.f + __expf(-x));
        const int m = map[row] - 1;
        float t = 0.f, w;
        if (m >= 0 && labels[plan[2 * M + m]] == c) { t = iou[m]; w = t; }
        else w = alpha * (gamma == 2.f ? p * p : __powf(p, gamma));
        const float bce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
        acc += w * bce;
        store_f(grad + e, s_vfl * w * (p - t));
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, s_vfl * s);
}

// ---------------------------------------------------------------------------------------------
// w[b,q] = matched ? iou : sigmoid(max_c teacher_logits)   (computed by each of the row's four edge threads)
// `edge`: the caller is one of FOUR adjacent lanes working on the row (its four box edges): each scans a quarter of the class
// logits and two cross-lane maxima finish the job (every lane scanning all C logits of its row made the teacher-logit reads -
// 80 strided 2-byte loads per thread - the longest part of head_phase2_kernel)
template <typename T>
__device__ __forceinline__ float row_weight(const T *__restrict__ tlogits, View tv, int m /* pair index or -1 */,
                                            const float *__restrict__ iou, int b, int q, int C, int edge) {
    if (m >= 0) return iou[m];                             // (uniform over the row's four lanes)
    const T *p = tlogits + (int64_t)b * tv.sb + (int64_t)q * tv.sq;
    float mx = -INFINITY;
    for (int c = edge; c < C; c += 4) mx = fmaxf(mx, load_f(p + c));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    return 1.f / (1.f + __expf(-mx));
}

// one thread per (b, q, edge) row of NB bins
template <typename T, int NB>
__device__ __forceinline__ void ddf_body(int blk, float *red, const T *__restrict__ pred, View pv,
                                         const T *__restrict__ teacher, View tv,
                                         const int *__restrict__ map, const T *__restrict__ tlogits, View tlv,
                                         const float *__restrict__ iou, int B, int Q, int C,
                                         float temp, float c_pos, float c_neg,
                                         T *__restrict__ grad /* [B,Q,4*NB] */,
                                         float *__restrict__ out) {
    const int r = blk * kLT + threadIdx.x;
    float loss = 0.f;
    if (r < B * Q * 4) {
        const int row = r >> 2, edge = r & 3;
        const int b = row / Q, q = row - b * Q;
        const T *pp = pred + (int64_t)b * pv.sb + (int64_t)q * pv.sq + edge * NB;
        const T *tp = teacher + (int64_t)b * tv.sb + (int64_t)q * tv.sq + edge * NB;
        float pm = -INFINITY, tm = -INFINITY;
        float pl[NB], tl[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            pl[j] = load_f(pp + j) / temp; tl[j] = load_f(tp + j) / temp;
            pm = fmaxf(pm, pl[j]); tm = fmaxf(tm, tl[j]);
        }
        float ps = 0.f, ts = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) { ps += __expf(pl[j] - pm); ts += __expf(tl[j] - tm); }
        const float plz = pm + __logf(ps), tlz = tm + __logf(ts);
        const int mrow = map[row] - 1;
        const bool pos = mrow >= 0;
        const float coef = (pos ? c_pos : c_neg) * row_weight(tlogits, tlv, mrow, iou, b, q, C, edge);
        float kl = 0.f;
        T *gp = grad + ((int64_t)row * 4 + edge) * NB;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float lq = pl[j] - plz, lt = tl[j] - tlz;
            const float tj = __expf(lt);
            kl += tj > 0.f ? tj * (lt - lq) : 0.f;
            store_f(gp + j, coef * temp * (__expf(lq) - tj));
        }
        loss = coef * temp * temp * kl;
    }
    const float s = block_sum(loss, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, s);
}

struct FglTable { float w[64]; int reg_max; float reg_scale; };

// one thread per (pair, edge); ADDS its gradient onto grad (written by ddf_kernel or zeroed)
template <typename T, int NB>
__device__ __forceinline__ void fgl_body(int blk, float *red, const T *__restrict__ pred, View pv,
                                         const float *__restrict__ ref, View rv,
                                         const float *__restrict__ tgt_boxes,
                                         const int64_t *__restrict__ plan, int M, int count, int Q,
                                         const float *__restrict__ iou, const FglTable &tab,
                                         float s_fgl, T *__restrict__ grad,
                                         float *__restrict__ out) {
    const int r = blk * kLT + threadIdx.x;
    float loss = 0.f;
    if (r < count * 4) {
        const int m = r >> 2, edge = r & 3;
        const int64_t b = plan[m], q = plan[M + m], t = plan[2 * (int64_t)M + m];
        const float *rp = ref + b * rv.sb + q * rv.sq;
        const float px = rp[0], py = rp[1], pw = rp[2], ph = rp[3];
        const float4 tb = *reinterpret_cast<const float4 *>(tgt_boxes + t * 4);
        const float tw = 0.5f * fmaxf(tb.z, 0.f), th = 0.5f * fmaxf(tb.w, 0.f);
        const float rs = fabsf(tab.reg_scale);
        // bbox2distance (arch/utils.py:328-354)
        float d;
        if (edge == 0) d = (px - (tb.x - tw)) / (pw / rs + 1e-16f) - 0.5f * rs;
        else if (edge == 1) d = (py - (tb.y - th)) / (ph / rs + 1e-16f) - 0.5f * rs;
        else if (edge == 2) d = ((tb.x + tw) - px) / (pw / rs + 1e-16f) - 0.5f * rs;
        else d = ((tb.y + th) - py) / (ph / rs + 1e-16f) - 0.5f * rs;
        // translate_gt (arch/utils.py:267-325)
        int cnt = 0;
        for (int j = 0; j <= tab.reg_max; ++j) cnt += (tab.w[j] - d) <= 0.f ? 1 : 0;
        const int idx = cnt - 1;
        float fidx, wr, wl;
        if (idx < 0) { fidx = 0.f; wr = 0.f; wl = 1.f; }
        else if (idx >= tab.reg_max) { fidx = (float)tab.reg_max - 0.1f; wr = 1.f; wl = 0.f; }
        else {
            const float dl = fabsf(d - tab.w[idx]), dr = fabsf(tab.w[idx + 1] - d);
            wr = dl / (dl + dr); wl = 1.f - wr; fidx = (float)idx;
        }
        fidx = fminf(fmaxf(fidx, 0.f), (float)tab.reg_max - 0.1f);
        const int left = (int)fidx;
        // two-bin cross entropy (dfine_criterion.py:837-858)
        const T *pp = pred + b * pv.sb + q * pv.sq + edge * NB;
        float x[NB], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NB; ++j) { x[j] = load_f(pp + j); mx = fmaxf(mx, x[j]); }
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) se += __expf(x[j] - mx);
        const float lz = mx + __logf(se);
        float xl = 0.f, xr = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) { xl = j == left ? x[j] : xl; xr = j == left + 1 ? x[j] : xr; }
        const float wi = s_fgl * iou[m];
        loss = wi * (wl * (lz - xl) + wr * (lz - xr));
        T *gp = grad + ((b * Q + q) * 4 + edge) * NB;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float g = wi * ((wl + wr) * __expf(x[j] - lz) - (j == left ? wl : 0.f) - (j == left + 1 ? wr : 0.f));
            store_f(gp + j, load_f(gp + j) + g);
        }
    }
    const float s = block_sum(loss, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, s);
}

// Second phase of a head (after pair_box_kernel): varifocal loss, DDF and FGL are independent of each other - ONE launch whose
// blocks take the three roles (they were 3-4 launches of 8-30 us each, back to back on one stream).
template <typename T>
struct Phase2 {
    const T *logits; View lv; const int *map_cls; const int64_t *cls_plan; int M_cls; const int64_t *labels; const float *iou_cls;
    float alpha, gamma, s_vfl; T *grad_logits;
    const T *corners; View pv; const T *teacher; View tcv; const int *map_box; const T *tlogits; View tlv; const float *iou_box;
    float temp, c_pos, c_neg; T *grad_ddf;
    const float *ref; View rv; const float *tgt_boxes; const int64_t *box_plan; int M_box; float s_fgl; T *grad_fgl;
    float *out; int B, Q, C, nb_vfl, nb_ddf;
    const float *dev_scales; const int *dev_m_box;        // device-resident normalisers / box-plan length (or null)
    FglTable tab;
};

template <typename T>
__global__ __launch_bounds__(kLT) void head_phase2_kernel(const Phase2<T> a) {
    __shared__ float red[kLT / 64];
    const int blk = blockIdx.x;
    const float *ds = a.dev_scales;
    if (blk < a.nb_vfl)
        vfl_body<T>(blk, a.nb_vfl, red, a.logits, a.lv, a.map_cls, a.cls_plan, a.M_cls, a.labels, a.iou_cls, a.B, a.Q, a.C, a.alpha,
                    a.gamma, ds ? ds[0] : a.s_vfl, a.grad_logits, a.out);
    else if (blk < a.nb_vfl + a.nb_ddf)
        ddf_body<T, 33>(blk - a.nb_vfl, red, a.corners, a.pv, a.teacher, a.tcv, a.map_box, a.tlogits, a.tlv, a.iou_box, a.B, a.Q, a.C,
                        a.temp, ds ? ds[4] : a.c_pos, ds ? ds[5] : a.c_neg, a.grad_ddf, a.out + 4);
    else
        fgl_body<T, 33>(blk - a.nb_vfl - a.nb_ddf, red, a.corners, a.pv, a.ref, a.rv, a.tgt_boxes, a.box_plan, a.M_box,
                        a.dev_m_box ? *a.dev_m_box : a.M_box, a.Q, a.iou_box, a.tab, ds ? ds[3] : a.s_fgl, a.grad_fgl, a.out + 3);
}

// Backward of the fused head losses: the closed-form gradients of the forward pass scaled by the upstream gradient g[5] of
// the (vfl, l1, giou, fgl, ddf) vector, in place and in ONE launch (the torch composition was 5 multiplies, 2 adds and the
// 0-d casts: ~8 launches per head, 11 heads, in the host-bound stretch between the matcher sync and the decoder's backward):
//   grad_logits *= g[0];   grad_l1 = grad_l1 * g[1] + grad_giou * g[2];   grad_fgl = grad_fgl * g[3] + grad_ddf * g[4]
template <typename T>
__global__ __launch_bounds__(256) void head_grads_scale_kernel(const float *__restrict__ g, T *__restrict__ gl, int64_t nl,
                                                               float *__restrict__ gb, const float *__restrict__ gg, int64_t nb,
                                                               T *__restrict__ gf, const T *__restrict__ gd, int64_t nc) {
    const float g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
    const int64_t step = (int64_t)gridDim.x * 256, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t i = t * 4; i + 3 < nl; i += step * 4) {
        const f32x4 v = Vec4<T>::load(gl + i);
        Vec4<T>::store(gl + i, {v.x * g0, v.y * g0, v.z * g0, v.w * g0});
    }
    for (int64_t i = (nl & ~(int64_t)3) + t; i < nl; i += step) store_f(gl + i, load_f(gl + i) * g0);
    for (int64_t i = t; i < nb; i += step) gb[i] = gb[i] * g1 + gg[i] * g2;
    if (gd) {
        for (int64_t i = t * 4; i + 3 < nc; i += step * 4) {
            const f32x4 a = Vec4<T>::load(gf + i), b = Vec4<T>::load(gd + i);
            Vec4<T>::store(gf + i, {a.x * g3 + b.x * g4, a.y * g3 + b.y * g4, a.z * g3 + b.z * g4, a.w * g3 + b.w * g4});
        }
        for (int64_t i = (nc & ~(int64_t)3) + t; i < nc; i += step) store_f(gf + i, load_f(gf + i) * g3 + load_f(gd + i) * g4);
    } else {
        for (int64_t i = t * 4; i + 3 < nc; i += step * 4) {
            const f32x4 a = Vec4<T>::load(gf + i);
            Vec4<T>::store(gf + i, {a.x * g3, a.y * g3, a.z * g3, a.w * g3});
        }
        for (int64_t i = (nc & ~(int64_t)3) + t; i < nc; i += step) store_f(gf + i, load_f(gf + i) * g3);
    }
}

}  // namespace dfine


using namespace dfine;

static thread_local bool g_head_prezeroed = false;

static int head_losses_impl(
    const void *logits, int64_t l_sb, int64_t l_sq, const float *boxes, int64_t b_sb, int64_t b_sq,
    const void *corners, int64_t c_sb, int64_t c_sq, const float *ref, int64_t r_sb, int64_t r_sq,
    const void *teacher_corners, int64_t tc_sb, int64_t tc_sq, const void *teacher_logits,
    int64_t tl_sb, int64_t tl_sq, const int64_t *cls_plan, int M_cls, const int64_t *box_plan,
    int M_box, const int64_t *tgt_labels, const float *tgt_boxes, const float *wtable, int reg_max,
    float reg_scale, float alpha, float gamma, float temp, float s_vfl, float s_l1, float s_giou,
    float s_fgl, float ddf_c_pos, float ddf_c_neg,
    void *grad_logits, float *grad_l1, float *grad_giou, void *grad_corners_fgl,
    void *grad_corners_ddf, float *iou_cls, float *iou_box, int *map_cls, int *map_box, float *wrow,
    float *out, int dtype, int B, int Q, int C, const float *dev_scales, const int *dev_m_box, void *stream);

extern "C" {

// out[5] = {vfl, l1, giou, fgl, ddf} sums (zeroed here).  See include/dfine_hip.h.
int dfine_head_losses(
    const void *logits, int64_t l_sb, int64_t l_sq, const float *boxes, int64_t b_sb, int64_t b_sq,
    const void *corners, int64_t c_sb, int64_t c_sq, const float *ref, int64_t r_sb, int64_t r_sq,
    const void *teacher_corners, int64_t tc_sb, int64_t tc_sq, const void *teacher_logits,
    int64_t tl_sb, int64_t tl_sq, const int64_t *cls_plan, int M_cls, const int64_t *box_plan,
    int M_box, const int64_t *tgt_labels, const float *tgt_boxes, const float *wtable, int reg_max,
    float reg_scale, float alpha, float gamma, float temp, float s_vfl, float s_l1, float s_giou,
    float s_fgl, float ddf_c_pos, float ddf_c_neg,
    void *grad_logits, float *grad_l1, float *grad_giou, void *grad_corners_fgl,
    void *grad_corners_ddf, float *iou_cls, float *iou_box, int *map_cls, int *map_box, float *wrow,
    float *out, int dtype, int B, int Q, int C, void *stream) {
    return head_losses_impl(logits, l_sb, l_sq, boxes, b_sb, b_sq, corners, c_sb, c_sq, ref, r_sb, r_sq, teacher_corners, tc_sb, tc_sq,
                            teacher_logits, tl_sb, tl_sq, cls_plan, M_cls, box_plan, M_box, tgt_labels, tgt_boxes, wtable, reg_max,
                            reg_scale, alpha, gamma, temp, s_vfl, s_l1, s_giou, s_fgl, ddf_c_pos, ddf_c_neg, grad_logits, grad_l1,
                            grad_giou, grad_corners_fgl, grad_corners_ddf, iou_cls, iou_box, map_cls, map_box, wrow, out, dtype, B, Q,
                            C, nullptr, nullptr, stream);
}

// The same launch group with the six scalar factors (s_vfl, s_l1, s_giou, s_fgl, ddf_c_pos, ddf_c_neg) read from DEVICE memory
// (`scales`, written by dfine_criterion_scales) and, when `box_count` is given, the number of valid entries of the box plan
// read from device memory too (M_box is then the plan's row stride / capacity).
int dfine_head_losses_dev(
    const void *logits, int64_t l_sb, int64_t l_sq, const float *boxes, int64_t b_sb, int64_t b_sq,
    const void *corners, int64_t c_sb, int64_t c_sq, const float *ref, int64_t r_sb, int64_t r_sq,
    const void *teacher_corners, int64_t tc_sb, int64_t tc_sq, const void *teacher_logits,
    int64_t tl_sb, int64_t tl_sq, const int64_t *cls_plan, int M_cls, const int64_t *box_plan,
    int M_box, const int64_t *tgt_labels, const float *tgt_boxes, const float *wtable, int reg_max,
    float reg_scale, float alpha, float gamma, float temp, const float *scales, const int *box_count,
    void *grad_logits, float *grad_l1, float *grad_giou, void *grad_corners_fgl,
    void *grad_corners_ddf, float *iou_cls, float *iou_box, int *map_cls, int *map_box, float *wrow,
    float *out, int dtype, int B, int Q, int C, void *stream) {
    if (!scales) return DFINE_E_BADARG;
    return head_losses_impl(logits, l_sb, l_sq, boxes, b_sb, b_sq, corners, c_sb, c_sq, ref, r_sb, r_sq, teacher_corners, tc_sb, tc_sq,
                            teacher_logits, tl_sb, tl_sq, cls_plan, M_cls, box_plan, M_box, tgt_labels, tgt_boxes, wtable, reg_max,
                            reg_scale, alpha, gamma, temp, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, grad_logits, grad_l1, grad_giou,
                            grad_corners_fgl, grad_corners_ddf, iou_cls, iou_box, map_cls, map_box, wrow, out, dtype, B, Q, C, scales,
                            box_count, stream);
}

}  // extern "C"

static int head_losses_impl(
    const void *logits, int64_t l_sb, int64_t l_sq, const float *boxes, int64_t b_sb, int64_t b_sq,
    const void *corners, int64_t c_sb, int64_t c_sq, const float *ref, int64_t r_sb, int64_t r_sq,
    const void *teacher_corners, int64_t tc_sb, int64_t tc_sq, const void *teacher_logits,
    int64_t tl_sb, int64_t tl_sq, const int64_t *cls_plan, int M_cls, const int64_t *box_plan,
    int M_box, const int64_t *tgt_labels, const float *tgt_boxes, const float *wtable, int reg_max,
    float reg_scale, float alpha, float gamma, float temp, float s_vfl, float s_l1, float s_giou,
    float s_fgl, float ddf_c_pos, float ddf_c_neg,
    void *grad_logits, float *grad_l1, float *grad_giou, void *grad_corners_fgl,
    void *grad_corners_ddf, float *iou_cls, float *iou_box, int *map_cls, int *map_box, float *wrow,
    float *out, int dtype, int B, int Q, int C, const float *dev_scales, const int *dev_m_box, void *stream) {
    if (!logits || !boxes || !out || !grad_logits || !grad_l1 || !grad_giou ||
        !map_cls || !map_box || B < 1 || Q < 1 || C < 1)
        return DFINE_E_BADARG;
    // a batch without targets has empty (null) target tensors and empty plans: every query is background
    if ((!tgt_boxes || !tgt_labels) && (M_cls > 0 || M_box > 0)) return DFINE_E_BADARG;
    if (dtype != DFINE_F32 && dtype != DFINE_BF16) return DFINE_E_BADARG;
    if (corners && (reg_max != 32 || !ref || !wtable || !grad_corners_fgl)) return DFINE_E_BADARG;
    if ((M_cls > 0 && (!cls_plan || !iou_cls)) || (M_box > 0 && (!box_plan || !iou_box))) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = dtype == DFINE_F32 ? 4 : 2;
    // the host wrapper lays [out(8) | grad_l1 | grad_giou | map_cls | map_box | (pad to 16 B) grad_corners_fgl] out back to back:
    // ONE fill instead of six (each fill is its own queue packet, 11 heads per step, all in the host-paced stretch after the
    // matcher sync); other layouts get their fills one by one
    const size_t nq = (size_t)B * Q;
    const size_t fgl_bytes = corners ? esz * nq * 4 * 33 : 0;
    char *const o8 = reinterpret_cast<char *>(out);
    const size_t maps_end = 32 + nq * 40;
    const bool packed = reinterpret_cast<char *>(grad_l1) == o8 + 32 && grad_giou == grad_l1 + nq * 4 &&
                        reinterpret_cast<char *>(map_cls) == o8 + 32 + nq * 32 && map_box == map_cls + nq &&
                        (!corners || reinterpret_cast<char *>(grad_corners_fgl) == o8 + (maps_end + 15) / 16 * 16);
    // one-shot request of the calling thread (dfine_head_losses_prezeroed_once): the packed block IS zero already - the caller cleared
    // the blocks of all heads of the step with one fill
    const bool prezeroed = g_head_prezeroed;
    g_head_prezeroed = false;
    if (prezeroed) {
        if (!packed) return DFINE_E_BADARG;
    } else if (packed) {
        zero_fill_async(out, corners ? (maps_end + 15) / 16 * 16 + fgl_bytes : maps_end, st);
    } else {
        zero_fill_async(out, 5 * sizeof(float), st);
        zero_fill_async(grad_l1, sizeof(float) * nq * 4, st);
        zero_fill_async(grad_giou, sizeof(float) * nq * 4, st);
        zero_fill_async(map_cls, sizeof(int) * nq, st);
        zero_fill_async(map_box, sizeof(int) * nq, st);
        if (corners) zero_fill_async(grad_corners_fgl, fgl_bytes, st);
    }
    const View bv{b_sb, b_sq};
    if (M_cls > 0 || M_box > 0) {   // IoU of the classification matching (VFL soft labels); L1 / GIoU of the box matching (+ IoU weights of FGL / DDF)
        const int nb_cls = (M_cls + kLT - 1) / kLT, nb_box = (M_box + kLT - 1) / kLT;
        hipLaunchKernelGGL(pair_box_kernel<float>, dim3(nb_cls + nb_box), dim3(kLT), 0, st, boxes, bv, tgt_boxes, cls_plan, M_cls,
                           iou_cls, map_cls, nb_cls, box_plan, M_box, iou_box, map_box, Q, grad_l1, grad_giou, out + 1, s_l1, s_giou,
                           dev_scales, dev_m_box);
    }
    const int64_t n = (int64_t)B * Q * C;
    if (n >= ((int64_t)1 << 31) - 2048 * kLT) return DFINE_E_BADARG;      // the kernels index the logits with 32 bits
    // (every block ends with ONE atomic onto the same loss scalar, and same-address atomics retire one after the other at L2:
    // 2 048 blocks made that tail the longest part of the launch)
    constexpr int vb_cap = 512;
    const int vb = (int)((n + kLT - 1) / kLT < vb_cap ? (n + kLT - 1) / kLT : vb_cap);
    if (corners && teacher_corners && (!teacher_logits || !grad_corners_ddf)) return DFINE_E_BADARG;
    const int nb_ddf = corners && teacher_corners ? (B * Q * 4 + kLT - 1) / kLT : 0;
    const int nb_fgl = corners && M_box > 0 ? (M_box * 4 + kLT - 1) / kLT : 0;
    auto launch = [&](auto tag) {
        using T = decltype(tag);
        Phase2<T> a{};
        a.logits = (const T *)logits; a.lv = View{l_sb, l_sq}; a.map_cls = map_cls; a.cls_plan = cls_plan; a.M_cls = M_cls;
        a.labels = tgt_labels; a.iou_cls = iou_cls; a.alpha = alpha; a.gamma = gamma; a.s_vfl = s_vfl; a.grad_logits = (T *)grad_logits;
        a.corners = (const T *)corners; a.pv = View{c_sb, c_sq}; a.teacher = (const T *)teacher_corners; a.tcv = View{tc_sb, tc_sq};
        a.map_box = map_box; a.tlogits = (const T *)teacher_logits; a.tlv = View{tl_sb, tl_sq}; a.iou_box = iou_box;
        a.temp = temp; a.c_pos = ddf_c_pos; a.c_neg = ddf_c_neg; a.grad_ddf = (T *)grad_corners_ddf;
        a.ref = ref; a.rv = View{r_sb, r_sq}; a.tgt_boxes = tgt_boxes; a.box_plan = box_plan; a.M_box = M_box; a.s_fgl = s_fgl;
        a.grad_fgl = (T *)grad_corners_fgl;
        a.out = out; a.B = B; a.Q = Q; a.C = C; a.nb_vfl = vb; a.nb_ddf = nb_ddf;
        a.dev_scales = dev_scales; a.dev_m_box = dev_m_box;
        if (nb_fgl) {
            for (int j = 0; j <= reg_max; ++j) a.tab.w[j] = wtable[j];
            a.tab.reg_max = reg_max; a.tab.reg_scale = reg_scale;
        }
        hipLaunchKernelGGL(head_phase2_kernel<T>, dim3(vb + nb_ddf + nb_fgl), dim3(kLT), 0, st, a);
    };
    if (dtype == DFINE_F32) launch(float{});
    else launch(uint16_t{});
    (void)wrow;          // (the DDF row weights are computed in place by the edge threads; the scratch argument stays in the ABI)
    return check_launch();
}

extern "C" {

// One-shot: the NEXT dfine_head_losses / dfine_head_losses_dev call of the calling thread finds its packed output block
// [out(8) | grad_l1 | grad_giou | map_cls | map_box | pad | grad_corners_fgl] zero-filled by the caller and skips its own fill
// (11 heads per training step: one fill of a common arena instead of 11).  A call whose buffers are not packed that way fails.
int dfine_head_losses_prezeroed_once(void) {
    g_head_prezeroed = true;
    return DFINE_OK;
}

int dfine_head_grads_scale(const float *g, void *grad_logits, int64_t n_logits, float *grad_l1, const float *grad_giou,
                           int64_t n_box, void *grad_corners_fgl, const void *grad_corners_ddf, int64_t n_corners, int dtype,
                           void *stream) {
    if (!g || !grad_logits || !grad_l1 || !grad_giou || n_logits < 0 || n_box < 0 || n_corners < 0 ||
        (n_corners > 0 && !grad_corners_fgl) || (dtype != DFINE_F32 && dtype != DFINE_BF16))
        return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (n_logits + n_corners) / 4 + n_box;
    if (work == 0 && n_logits == 0 && n_corners == 0) return DFINE_OK;
    const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (work + 255) / 256));
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(head_grads_scale_kernel<float>, dim3(grid), dim3(256), 0, st, g, (float *)grad_logits, n_logits,
                           grad_l1, grad_giou, n_box, (float *)grad_corners_fgl, (const float *)grad_corners_ddf, n_corners);
    else
        hipLaunchKernelGGL(head_grads_scale_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, g, (uint16_t *)grad_logits, n_logits,
                           grad_l1, grad_giou, n_box, (uint16_t *)grad_corners_fgl, (const uint16_t *)grad_corners_ddf, n_corners);
    return check_launch();
}

}  // extern "C"
