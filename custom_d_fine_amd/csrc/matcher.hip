// A11 + A12 - Hungarian matcher on the device: per-image cost blocks + linear sum assignment.
//
// Reference: HungarianMatcher.forward (src/d_fine/matcher.py:110-257) computes the dense
// [B*Q, sum T] fp32 cost, copies it to the host and calls scipy.optimize.
// linear_sum_assignment per image.  Here
//   match_cost_kernel : only the block-diagonal [T_b, Q] blocks, fp32, the reference's operation
//                       order (sigmoid -> focal pos/neg -> L1 -> GIoU -> weighted sum -> NaN clean-up);
//   lsap_kernel       : one 64-lane wave per (head, image) problem runs SciPy's rectangular LSAP
//                       (Crouse's shortest-augmenting-path, scipy 1.15 rectangular_lsap.cpp) in
//                       float64 with all state in LDS.  The column scan of every path step is
//                       spread over the 64 lanes and reduced with wave shuffles using a
//                       combine rule that reproduces the sequential scan's tie-breaking exactly
//                       (strict '<', ties go to a still-unassigned column, otherwise to the first
//                       hit; the unscanned-column list starts reversed and shrinks by
//                       swap-with-last), so the indices are bit-identical to SciPy's on the same
//                       cost matrix.  Compiled with -ffp-contract=off (see build flags).
// Cost blocks are stored target-major, [K, B, Tmax, Q]: a row of the (transposed, T <= Q)
// assignment problem is one contiguous run of Q floats.
#include "common.h"

namespace dfine {

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void match_cost_kernel(
    const float *__restrict__ logits, const float *__restrict__ boxes,
    const int64_t *__restrict__ tgt_labels, const float *__restrict__ tgt_boxes,
    const int *__restrict__ tgt_offset, const float *__restrict__ extra, float *__restrict__ cost,
    int B, int Q, int C, int Tmax, float w_class, float w_bbox, float w_giou, float alpha,
    float gamma) {
    const int kb = blockIdx.y, b = kb % B;
    const int t0 = tgt_offset[b], nt = tgt_offset[b + 1] - t0;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nt * Q) return;
    const int t = e / Q, q = e - t * Q;

    const int64_t label = tgt_labels[t0 + t];
    const float x = logits[((int64_t)kb * Q + q) * C + label];
    const float p = 1.f / (1.f + expf(-x));                     // torch.sigmoid
    const bool sq = gamma == 2.0f;                              // torch.pow(x, 2) == x*x
    const float pg = sq ? p * p : powf(p, gamma);
    const float qg = sq ? (1.f - p) * (1.f - p) : powf(1.f - p, gamma);
    const float neg = (1.f - alpha) * pg * (-logf(1.f - p + 1e-8f));
    const float pos = alpha * qg * (-logf(p + 1e-8f));
    const float c_class = pos - neg;

    const float4 ob = *reinterpret_cast<const float4 *>(boxes + ((int64_t)kb * Q + q) * 4);
    const float4 tb = *reinterpret_cast<const float4 *>(tgt_boxes + (int64_t)(t0 + t) * 4);
    const float c_bbox = fabsf(ob.x - tb.x) + fabsf(ob.y - tb.y) + fabsf(ob.z - tb.z) + fabsf(ob.w - tb.w);

    // box_cxcywh_to_xyxy (w, h clamped at 0) + generalized_box_iou
    const float ow = 0.5f * fmaxf(ob.z, 0.f), oh = 0.5f * fmaxf(ob.w, 0.f);
    const float tw = 0.5f * fmaxf(tb.z, 0.f), th = 0.5f * fmaxf(tb.w, 0.f);
    const float ax0 = ob.x - ow, ay0 = ob.y - oh, ax1 = ob.x + ow, ay1 = ob.y + oh;
    const float bx0 = tb.x - tw, by0 = tb.y - th, bx1 = tb.x + tw, by1 = tb.y + th;
    const float area_a = (ax1 - ax0) * (ay1 - ay0), area_b = (bx1 - bx0) * (by1 - by0);
    const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
    const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
    const float inter = iw * ih;
    const float uni = area_a + area_b - inter;
    const float iou = inter / uni;
    const float hw = fmaxf(fmaxf(ax1, bx1) - fminf(ax0, bx0), 0.f);
    const float hh = fmaxf(fmaxf(ay1, by1) - fminf(ay0, by0), 0.f);
    const float hull = hw * hh;
    const float giou = iou - (hull - uni) / hull;

    float c = w_bbox * c_bbox + w_class * c_class + w_giou * (-giou);
    const int64_t o = ((int64_t)kb * Tmax + t) * Q + q;
    if (extra) c += extra[o];
    // torch.nan_to_num(C, nan=1.0): NaN -> 1, +-inf -> +-FLT_MAX
    if (c != c) c = 1.0f;
    else if (c == INFINITY) c = 3.402823466e+38f;
    else if (c == -INFINITY) c = -3.402823466e+38f;
    cost[o] = c;
}

// ---------------------------------------------------------------------------------------------
struct Cand {
    double m;     // minimum shortest-path cost seen
    int first;    // smallest list position with that cost
    int last_u;   // largest list position with that cost whose column is unassigned, or -1
};

__device__ __forceinline__ Cand combine(const Cand &a, const Cand &b) {
    if (a.m < b.m) return a;
    if (b.m < a.m) return b;
    Cand r;
    r.m = a.m;
    r.first = min(a.first, b.first);
    r.last_u = max(a.last_u, b.last_u);
    return r;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// One wave per problem.  `cost` block is [Tmax, Q] (target-major).  The assignment problem is
// solved on the n_small x n_large matrix (rows = the smaller side, SciPy's transpose rule).
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ cost,
                                                  const int *__restrict__ tgt_offset,
                                                  int *__restrict__ match_out, int B, int Q,
                                                  int Tmax, int T_total, int n_large_max,
                                                  int n_small_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int kb = blockIdx.x, b = kb % B, k = kb / B;
    const int t0 = tgt_offset[b], nt = tgt_offset[b + 1] - t0;
    const int lane = threadIdx.x;
    int *out = match_out + (int64_t)k * T_total + t0;
    if (nt == 0) return;
    const float *cb = cost + (int64_t)kb * Tmax * Q;

    const bool rows_are_targets = nt <= Q;      // SciPy transposes a tall [Q, T] matrix (Q > T)
    const int nr = rows_are_targets ? nt : Q;   // small side
    const int nc = rows_are_targets ? Q : nt;   // large side
    // element (row i, col j) of the solved matrix
    auto C = [&](int i, int j) -> double {
        return rows_are_targets ? (double)cb[(int64_t)i * Q + j] : (double)cb[(int64_t)j * Q + i];
    };

    double *u = reinterpret_cast<double *>(lds_raw);                 // [n_small]
    double *v = u + n_small_max;                                     // [n_large]
    double *spc = v + n_large_max;                                   // [n_large]
    int *path = reinterpret_cast<int *>(spc + n_large_max);          // [n_large]
    int *row4col = path + n_large_max;                               // [n_large]
    int *remaining = row4col + n_large_max;                          // [n_large]
    int *col4row = remaining + n_large_max;                          // [n_small]
    unsigned char *SC = reinterpret_cast<unsigned char *>(col4row + n_small_max);   // [n_large]
    unsigned char *SR = SC + n_large_max;                                           // [n_small]

    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    __syncthreads();

    bool feasible = true;
    for (int cur = 0; cur < nr && feasible; ++cur) {
        // ---- shortest augmenting path from row `cur`
        for (int j = lane; j < nc; j += 64) { remaining[j] = nc - j - 1; SC[j] = 0; spc[j] = INFINITY; }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        __syncthreads();
        double min_val = 0.0;
        int n_rem = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            Cand best = {INFINITY, 0x7fffffff, -1};
            for (int it = lane; it < n_rem; it += 64) {
                const int j = remaining[it];
                const double r = min_val + C(i, j) - ui - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const Cand c = {s, it, row4col[j] == -1 ? it : -1};
                best = combine(best, c);      // `it` increases along a lane: same rule applies
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                Cand o;
                o.m = shfl_xor_f64(best.m, m);
                o.first = __shfl_xor(best.first, m, 64);
                o.last_u = __shfl_xor(best.last_u, m, 64);
                best = combine(best, o);
            }
            min_val = best.m;
            if (min_val == INFINITY) { feasible = false; break; }
            const int index = best.last_u >= 0 ? best.last_u : best.first;
            const int j = remaining[index];
            const int r4c = row4col[j];
            if (r4c == -1) sink = j; else i = r4c;
            __syncthreads();                   // everyone has read remaining[index] / row4col[j]
            if (lane == 0) { SC[j] = 1; remaining[index] = remaining[n_rem - 1]; }
            --n_rem;
            __syncthreads();
        }
        if (!feasible) break;
        // ---- dual updates
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += min_val;
            else if (SR[r]) u[r] += min_val - spc[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= min_val - spc[j];
        __syncthreads();
        // ---- augment along the path
        if (lane == 0) {
            int j = sink;
            for (;;) {
                const int r = path[j];
                row4col[j] = r;
                const int tmp = col4row[r];
                col4row[r] = j;
                j = tmp;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }

    if (rows_are_targets) {
        for (int t = lane; t < nt; t += 64) out[t] = feasible ? col4row[t] : -1;
    } else {
        for (int t = lane; t < nt; t += 64) out[t] = feasible ? row4col[t] : -1;
    }
}

static size_t lsap_lds_bytes(int n_large, int n_small) {
    size_t b = sizeof(double) * ((size_t)n_small + 2 * (size_t)n_large) +
               sizeof(int) * (3 * (size_t)n_large + (size_t)n_small) + (size_t)n_large + (size_t)n_small;
    return (b + 15) & ~(size_t)15;
}

static int launch_lsap(const float *cost, const int *tgt_offset, int *match_out, int K, int B, int Q,
                       int Tmax, int T_total, hipStream_t st) {
    const int n_large = Q > Tmax ? Q : Tmax, n_small = Q > Tmax ? Tmax : Q;
    const size_t lds = lsap_lds_bytes(n_large, n_small);
    if (lds > 150 * 1024) return DFINE_E_BADARG;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
    }
    hipLaunchKernelGGL(lsap_kernel, dim3(K * B), dim3(64), lds, st, cost, tgt_offset, match_out, B,
                       Q, Tmax, T_total, n_large, n_small);
    return check_launch();
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_match_ws_bytes(int K, int B, int Q, int Tmax) {
    (void)K; (void)B; (void)Q; (void)Tmax;
    return 16;  // all LSAP state lives in LDS; kept in the ABI for a future global-memory path
}

int dfine_match(const float *logits, const float *boxes, const int64_t *tgt_labels,
                const float *tgt_boxes, const int *tgt_offset, const float *extra_cost,
                float *cost_out, void *lsap_ws, int *match_out, int K, int B, int Q, int C, int Tmax,
                int T_total, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                void *stream) {
    (void)lsap_ws;
    if (K < 1 || B < 1 || Q < 1 || C < 1 || Tmax < 0 || T_total < 0) return DFINE_E_BADARG;
    if (Tmax == 0 || T_total == 0) return DFINE_OK;
    if (!logits || !boxes || !tgt_labels || !tgt_boxes || !tgt_offset || !cost_out || !match_out)
        return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int per = Tmax * Q;
    hipLaunchKernelGGL(match_cost_kernel, dim3((per + 255) / 256, K * B), dim3(256), 0, st, logits,
                       boxes, tgt_labels, tgt_boxes, tgt_offset, extra_cost, cost_out, B, Q, C, Tmax,
                       w_class, w_bbox, w_giou, alpha, gamma);
    if (int e = check_launch()) return e;
    return launch_lsap(cost_out, tgt_offset, match_out, K, B, Q, Tmax, T_total, st);
}

int dfine_lsap(const float *cost, const int *tgt_offset, void *lsap_ws, int *match_out, int K, int B,
               int Q, int Tmax, int T_total, void *stream) {
    (void)lsap_ws;
    if (K < 1 || B < 1 || Q < 1 || Tmax < 0 || T_total < 0) return DFINE_E_BADARG;
    if (Tmax == 0 || T_total == 0) return DFINE_OK;
    if (!cost || !tgt_offset || !match_out) return DFINE_E_BADARG;
    return launch_lsap(cost, tgt_offset, match_out, K, B, Q, Tmax, T_total, (hipStream_t)stream);
}

}  // extern "C"
