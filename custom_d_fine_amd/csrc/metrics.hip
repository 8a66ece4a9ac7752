// (f2) Evaluation hand-off, instance masks: the pairwise mask IoU of the reference's Validator
// (src/dl/validator.py:283-293: pm.float().flatten(1) @ gm.float().flatten(1).T, areas, union, inter / union) on
// BIT-PACKED masks.  A binary mask is 1 bit per pixel here (the reference keeps uint8 maps and, to survive a validation set,
// round-trips them through pycocotools RLE on the host - src/dl/utils.py:1040-1160); the intersection of two masks is
// popcount(a & b) over 64-bit words.  Counts are exact integers, and inter / union is one IEEE fp32 division of two
// exactly representable numbers (H * W < 2^24), so the values are bit-identical to the reference's fp32 matmul route.
//   mask_pack_bits_kernel  [N, HW] uint8 (non-zero) / f32 / bf16 (> thresh)  ->  [N, W64] uint64.  A lane reads four
//                          consecutive pixels, a wave 256; ballot k collects pixel 4 * lane + k of every lane.  The bit
//                          order inside a 256-pixel chunk is therefore a fixed permutation of the pixels - the same for
//                          every mask, which is all popcount(a & b) needs.
//   mask_iou_bits_kernel   one wave per (prediction, ground truth) pair: lanes stride over the words, wave reduction.
#include "common.h"

namespace dfine {

template <int DT>   // 0: uint8 != 0, 1: f32 > thresh, 2: bf16 > thresh
__global__ __launch_bounds__(256) void mask_pack_bits_kernel(const void *__restrict__ masks, float thresh, int64_t HW, int64_t W64,
                                                            uint64_t *__restrict__ bits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = blockIdx.y;
    const int64_t chunks = W64 / 4;
    for (int64_t ch = (int64_t)blockIdx.x * 4 + wave; ch < chunks; ch += (int64_t)gridDim.x * 4) {
        const int64_t p0 = ch * 256 + 4 * lane;
        bool on[4] = {false, false, false, false};
        if (p0 + 3 < HW && (HW & 3) == 0) {                   // rows of masks stay 4-pixel aligned: one vector load
            if (DT == 0) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(masks) + n * HW + p0);
#pragma unroll
                for (int k = 0; k < 4; ++k) on[k] = ((v >> (8 * k)) & 0xffu) != 0;
            } else if (DT == 1) {
                const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(masks) + n * HW + p0);
                on[0] = v.x > thresh; on[1] = v.y > thresh; on[2] = v.z > thresh; on[3] = v.w > thresh;
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(masks) + n * HW + p0);
                on[0] = __uint_as_float(v.x << 16) > thresh; on[1] = __uint_as_float(v.x & 0xffff0000u) > thresh;
                on[2] = __uint_as_float(v.y << 16) > thresh; on[3] = __uint_as_float(v.y & 0xffff0000u) > thresh;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (p0 + k >= HW) continue;
                if (DT == 0) on[k] = reinterpret_cast<const uint8_t *>(masks)[n * HW + p0 + k] != 0;
                else if (DT == 1) on[k] = reinterpret_cast<const float *>(masks)[n * HW + p0 + k] > thresh;
                else on[k] = bf16_to_f32(reinterpret_cast<const uint16_t *>(masks)[n * HW + p0 + k]) > thresh;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t b = __ballot(on[k]);
            if (lane == k) bits[n * W64 + ch * 4 + k] = b;
        }
    }
}

__global__ __launch_bounds__(256) void mask_iou_bits_kernel(const uint64_t *__restrict__ pb, const uint64_t *__restrict__ gb, int Np,
                                                           int Ng, int64_t W64, float *__restrict__ iou) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)Np * Ng) return;
    const int p = (int)(pair / Ng), g = (int)(pair - (int64_t)p * Ng);
    const uint64_t *a = pb + (int64_t)p * W64, *b = gb + (int64_t)g * W64;
    int inter = 0, ap = 0, ag = 0;
    for (int64_t w = lane; w < W64; w += 64) {
        const uint64_t x = a[w], y = b[w];
        inter += __popcll(x & y);
        ap += __popcll(x);
        ag += __popcll(y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        inter += __shfl_xor(inter, off);
        ap += __shfl_xor(ap, off);
        ag += __shfl_xor(ag, off);
    }
    if (lane == 0) {
        const int uni = ap + ag - inter;
        iou[pair] = uni > 0 ? __fdiv_rn((float)inter, (float)uni) : 0.f;
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_mask_bits_words(int64_t HW) { return 4 * ((HW + 255) / 256); }

int dfine_mask_pack_bits(const void *masks, int dtype, float thresh, int N, int64_t HW, void *bits, void *stream) {
    if (N == 0) return DFINE_OK;
    if (!masks || !bits || N < 0 || HW <= 0 || dtype < 0 || dtype > 2) return DFINE_E_BADARG;
    const int64_t W64 = dfine_mask_bits_words(HW);
    const int64_t chunks = W64 / 4;
    dim3 grid((unsigned)((chunks + 3) / 4 < 1024 ? (chunks + 3) / 4 : 1024), N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(mask_pack_bits_kernel<0>, grid, dim3(256), 0, st, masks, thresh, HW, W64, (uint64_t *)bits);
    else if (dtype == 1) hipLaunchKernelGGL(mask_pack_bits_kernel<1>, grid, dim3(256), 0, st, masks, thresh, HW, W64, (uint64_t *)bits);
    else hipLaunchKernelGGL(mask_pack_bits_kernel<2>, grid, dim3(256), 0, st, masks, thresh, HW, W64, (uint64_t *)bits);
    return check_launch();
}

int dfine_mask_iou_bits(const void *pred_bits, const void *gt_bits, int Np, int Ng, int64_t words, float *iou, void *stream) {
    if (Np == 0 || Ng == 0) return DFINE_OK;
    if (!pred_bits || !gt_bits || !iou || Np < 0 || Ng < 0 || words <= 0) return DFINE_E_BADARG;
    const int64_t pairs = (int64_t)Np * Ng;
    hipLaunchKernelGGL(mask_iou_bits_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint64_t *)pred_bits, (const uint64_t *)gt_bits, Np, Ng, words, iou);
    return check_launch();
}

}  // extern "C"
