// A13 - index bookkeeping of the set criterion ON THE DEVICE: gather plans of every matched head, the GO union of all
// matchings and the loss normalisers derived from its size, straight from the assignment vectors the matcher kernel left in
// device memory.  The step then has NO host <-> device synchronisation: the reference copies every matching to the host
// (src/d_fine/matcher.py:243-257), builds the union with torch.unique / torch.argsort per image
// (src/d_fine/dfine_criterion.py:570-591) and feeds its size back as Python numbers (:619-652); the round-2/3 build did the
// same once per step (one D2H copy, numpy, one plan upload) and the device idled ~4 ms per step behind that round trip.
//
//   go_image_kernel    one wave per image: the (query, target) pairs of all K heads -> sorted unique pairs + multiplicities
//                      (torch.unique(dim=0, return_counts=True): lexicographic), order by multiplicity exactly like
//                      torch.argsort(counts, descending=True) on the CPU (ATen sorts with std::sort = libstdc++ introsort,
//                      unstable but deterministic: restated step by step below, pinned against torch in tests/test_plans.py
//                      through oracle/np_ref.py::aten_argsort_desc), first pair per query wins, order of appearance kept
//   plan_compact_kernel  per-image results -> one [3, cap] plan (image-major, like the reference's list of per-image pairs),
//                      its length, and the [3, T] plan of every head (image, query, target row) in target order
//   criterion_scales_kernel  the scalar factors of every head-loss launch (weights / normalisers, DDF positive / negative
//                      balance) from the GO size, in the arithmetic of the reference's Python (double, float32 division of
//                      the clamp) - read by dfine_head_losses_dev from device memory
#include "common.h"

namespace dfine {

constexpr int kPlanMaxN = 4096;        // K * T_i pairs of one image
constexpr int kPlanMaxQ = 4096;

// ---- libstdc++ std::sort (bits/stl_algo.h, bits/stl_heap.h) on (value, index) arrays, comparator value_a > value_b ---------
struct KV { uint16_t *v; uint16_t *ix; };

__device__ __forceinline__ void kv_swap(KV a, int i, int j) {
    const uint16_t tv = a.v[i], ti = a.ix[i];
    a.v[i] = a.v[j]; a.ix[i] = a.ix[j];
    a.v[j] = tv; a.ix[j] = ti;
}

__device__ inline void unguarded_linear_insert(KV a, int last) {
    const uint16_t val = a.v[last], vi = a.ix[last];
    int next = last - 1;
    while (val > a.v[next]) {
        a.v[last] = a.v[next]; a.ix[last] = a.ix[next];
        last = next;
        --next;
    }
    a.v[last] = val; a.ix[last] = vi;
}

__device__ inline void insertion_sort(KV a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (a.v[i] > a.v[first]) {
            const uint16_t val = a.v[i], vi = a.ix[i];
            for (int j = i; j > first; --j) { a.v[j] = a.v[j - 1]; a.ix[j] = a.ix[j - 1]; }      // move_backward
            a.v[first] = val; a.ix[first] = vi;
        } else {
            unguarded_linear_insert(a, i);
        }
    }
}

__device__ inline void push_heap(KV a, int first, int hole, int top, uint16_t val, uint16_t vi) {
    int parent = (hole - 1) / 2;
    while (hole > top && a.v[first + parent] > val) {
        a.v[first + hole] = a.v[first + parent]; a.ix[first + hole] = a.ix[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.v[first + hole] = val; a.ix[first + hole] = vi;
}

__device__ inline void adjust_heap(KV a, int first, int hole, int len, uint16_t val, uint16_t vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (a.v[first + child] > a.v[first + child - 1]) --child;
        a.v[first + hole] = a.v[first + child]; a.ix[first + hole] = a.ix[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a.v[first + hole] = a.v[first + child - 1]; a.ix[first + hole] = a.ix[first + child - 1];
        hole = child - 1;
    }
    push_heap(a, first, hole, top, val, vi);
}

__device__ inline void heap_sort(KV a, int first, int last) {       // std::__partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; --parent) {
            adjust_heap(a, first, parent, len, a.v[first + parent], a.ix[first + parent]);
            if (parent == 0) break;
        }
    }
    while (last - first > 1) {
        --last;
        const uint16_t val = a.v[last], vi = a.ix[last];
        a.v[last] = a.v[first]; a.ix[last] = a.ix[first];
        adjust_heap(a, first, 0, last - first, val, vi);
    }
}

__device__ inline void move_median_to_first(KV a, int result, int x, int y, int z) {
    if (a.v[x] > a.v[y]) {
        if (a.v[y] > a.v[z]) kv_swap(a, result, y);
        else if (a.v[x] > a.v[z]) kv_swap(a, result, z);
        else kv_swap(a, result, x);
    } else if (a.v[x] > a.v[z]) kv_swap(a, result, x);
    else if (a.v[y] > a.v[z]) kv_swap(a, result, z);
    else kv_swap(a, result, y);
}

__device__ inline int unguarded_partition(KV a, int first, int last, int pivot) {
    while (true) {
        while (a.v[first] > a.v[pivot]) ++first;
        --last;
        while (a.v[pivot] > a.v[last]) --last;
        if (!(first < last)) return first;
        kv_swap(a, first, last);
        ++first;
    }
}

// std::sort(first, last, "value greater"): the recursion of __introsort_loop on the right part as an explicit stack (the
// two parts of a partition are disjoint, so the order in which they are finished does not matter)
__device__ inline void aten_sort_desc(KV a, int n) {
    if (n == 0) return;
    int stack[3 * 64], sp = 0;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    stack[sp++] = 0; stack[sp++] = n; stack[sp++] = 2 * lg;
    while (sp) {
        int depth = stack[--sp], last = stack[--sp];
        const int first = stack[--sp];
        while (last - first > 16) {
            if (depth == 0) { heap_sort(a, first, last); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            move_median_to_first(a, first, first + 1, mid, last - 1);
            const int cut = unguarded_partition(a, first + 1, last, first);
            stack[sp++] = cut; stack[sp++] = last; stack[sp++] = depth;
            last = cut;
        }
    }
    if (n > 16) {
        insertion_sort(a, 0, 16);
        for (int i = 16; i != n; ++i) unguarded_linear_insert(a, i);
    } else {
        insertion_sort(a, 0, n);
    }
}

// One wave per image.  cols [K, T] (query of target row t for head k), tgt_offset [B + 1].
// out: go_q / go_t [K * T] (image i writes at K * tgt_offset[i]), go_n [B].
__global__ __launch_bounds__(64) void go_image_kernel(const int *__restrict__ cols, const int *__restrict__ tgt_offset, int K, int T, int Q,
                                                      int *__restrict__ go_q, int *__restrict__ go_t, int *__restrict__ go_n) {
    __shared__ int key[kPlanMaxN], sorted[kPlanMaxN];
    __shared__ uint16_t cnt[kPlanMaxN], idx[kPlanMaxN];
    __shared__ uint32_t seen[kPlanMaxQ / 32];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int off = tgt_offset[img], Ti = tgt_offset[img + 1] - off;
    const int n = K * Ti;
    if (n == 0) {
        if (lane == 0) go_n[img] = 0;
        return;
    }
    for (int e = lane; e < n; e += 64) {
        const int k = e / Ti, j = e - k * Ti;
        key[e] = cols[(int64_t)k * T + off + j] * Ti + j;           // (query, target-in-image), lexicographic
    }
    for (int w = lane; w < (Q + 31) / 32; w += 64) seen[w] = 0u;
    __syncthreads();
    for (int e = lane; e < n; e += 64) {                             // rank sort (stable): n <= 4096, 64 lanes
        const int ke = key[e];
        int r = 0;
        for (int f = 0; f < n; ++f) {
            const int kf = key[f];
            r += (kf < ke) || (kf == ke && f < e) ? 1 : 0;
        }
        sorted[r] = ke;
    }
    __syncthreads();
    if (lane == 0) {
        int u = 0;
        for (int e = 0; e < n; ++e) {                                // unique pairs + multiplicities
            if (e == 0 || sorted[e] != sorted[e - 1]) { key[u] = sorted[e]; cnt[u] = 1; idx[u] = (uint16_t)u; ++u; }
            else ++cnt[u - 1];
        }
        aten_sort_desc(KV{cnt, idx}, u);                             // torch.argsort(counts, descending=True) on the CPU
        int c = 0;
        int *oq = go_q + (int64_t)K * off, *ot = go_t + (int64_t)K * off;
        for (int e = 0; e < u; ++e) {                                // the first pair of every query, in order of appearance
            const int kk = key[idx[e]];
            const int q = kk / Ti, t = kk - q * Ti;
            if (!(seen[q >> 5] >> (q & 31) & 1u)) {
                seen[q >> 5] |= 1u << (q & 31);
                oq[c] = q; ot[c] = t;
                ++c;
            }
        }
        go_n[img] = c;
    }
}

// head_plans int64 [K, 3, T]; go_plan int64 [3, cap]; go_count int32 [1] (+ [1] = float copy for a collective)
__global__ __launch_bounds__(256) void plan_compact_kernel(const int *__restrict__ cols, const int *__restrict__ tgt_offset, int K, int T, int B,
                                                           const int *__restrict__ go_q, const int *__restrict__ go_t,
                                                           const int *__restrict__ go_n, int64_t *__restrict__ head_plans,
                                                           int64_t *__restrict__ go_plan, int cap, int *__restrict__ go_count,
                                                           float *__restrict__ go_count_f) {
    __shared__ int start[1025];
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < B; ++i) { start[i] = s; s += go_n[i]; }
        start[B] = s;
        go_count[0] = s;
        if (go_count_f) go_count_f[0] = (float)s;
    }
    __syncthreads();
    for (int i = 0; i < B; ++i) {
        const int off = tgt_offset[i], n = start[i + 1] - start[i];
        const int *q = go_q + (int64_t)K * off, *t = go_t + (int64_t)K * off;
        for (int c = threadIdx.x; c < n; c += 256) {
            const int m = start[i] + c;
            go_plan[m] = i; go_plan[cap + m] = q[c]; go_plan[2 * (int64_t)cap + m] = off + t[c];
        }
    }
    for (int e = threadIdx.x; e < K * T; e += 256) {
        const int k = e / T, t = e - k * T;
        int lo = 0, hi = B;                                          // image of target row t: tgt_offset[lo] <= t < tgt_offset[lo + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (tgt_offset[mid] <= t) lo = mid; else hi = mid;
        }
        int64_t *p = head_plans + (int64_t)k * 3 * T;
        p[t] = lo; p[T + t] = cols[e]; p[2 * (int64_t)T + t] = t;
    }
}

// params: double [R, 12] = (s_vfl, go_box, n_box, w_bbox, w_giou, w_fgl, w_ddf, teacher, is_dn, rows_total, rows_pos_host, 8 / B)
// scales: float [R, 6] = (s_vfl, s_l1, s_giou, s_fgl, c_pos, c_neg)
__global__ void criterion_scales_kernel(const double *__restrict__ params, int R, const int *__restrict__ go_count,
                                        const float *__restrict__ go_sum, int world, float *__restrict__ scales) {
    if (threadIdx.x || blockIdx.x) return;
    const float tot = go_sum ? go_sum[0] : (float)go_count[0];
    const float nbg = fmaxf(tot / (float)world, 1.0f);               // torch.clamp(num_boxes_go / world, min=1).item()
    const double num_boxes_go = (double)nbg;
    const double rows_pos_go = 4.0 * (double)go_count[0];
    double num_pos = 0.0, num_neg = 0.0;
    for (int r = 0; r < R; ++r) {
        const double *p = params + 12 * r;
        const double n_box = p[1] != 0.0 ? num_boxes_go : p[2];
        float *s = scales + 6 * r;
        s[0] = (float)p[0];
        s[1] = (float)(p[3] / n_box);
        s[2] = (float)(p[4] / n_box);
        s[3] = (float)(p[5] / n_box);
        double c_pos = 0.0, c_neg = 0.0;
        if (p[7] != 0.0) {
            const double rows_pos = p[1] != 0.0 ? rows_pos_go : p[10];
            const double rows_neg = p[9] - rows_pos;
            if (p[8] == 0.0) {                                        // cached for the denoising heads (ref :223-229)
                num_pos = sqrt(rows_pos * p[11]);
                num_neg = sqrt(rows_neg * p[11]);
            }
            const double den = num_pos + num_neg;
            c_pos = rows_pos > 0.0 ? p[6] * num_pos / (den * rows_pos) : 0.0;
            c_neg = rows_neg > 0.0 ? p[6] * num_neg / (den * rows_neg) : 0.0;
        }
        s[4] = (float)c_pos;
        s[5] = (float)c_neg;
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_criterion_plans_supported(int K, int tmax, int Q) { return (int64_t)K * tmax <= kPlanMaxN && Q <= kPlanMaxQ && tmax <= Q; }

int64_t dfine_criterion_plans_ws_ints(int K, int T, int B) { return 2 * (int64_t)K * T + B; }

int dfine_criterion_plans(const int *cols, const int *tgt_offset, int K, int T, int B, int Q, int tmax, int64_t *head_plans,
                          int64_t *go_plan, int cap, int *go_count, float *go_count_f, int *ws, void *stream) {
    if (!cols || !tgt_offset || !head_plans || !go_plan || !go_count || !ws || K < 1 || T < 1 || B < 1 || B > 1024 ||
        cap < K * T || !dfine_criterion_plans_supported(K, tmax, Q))
        return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int *go_q = ws, *go_t = ws + (int64_t)K * T, *go_n = ws + 2 * (int64_t)K * T;
    hipLaunchKernelGGL(go_image_kernel, dim3(B), dim3(64), 0, st, cols, tgt_offset, K, T, Q, go_q, go_t, go_n);
    hipLaunchKernelGGL(plan_compact_kernel, dim3(1), dim3(256), 0, st, cols, tgt_offset, K, T, B, go_q, go_t, go_n, head_plans,
                       go_plan, cap, go_count, go_count_f);
    return check_launch();
}

int dfine_criterion_scales(const double *params, int R, const int *go_count, const float *go_sum, int world, float *scales,
                           void *stream) {
    if (!params || !go_count || !scales || R < 1 || world < 1) return DFINE_E_BADARG;
    hipLaunchKernelGGL(criterion_scales_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, params, R, go_count, go_sum, world, scales);
    return check_launch();
}

}  // extern "C"
