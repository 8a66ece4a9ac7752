// A4 - the contrastive-denoising query group of one training step (reference src/d_fine/arch/utils.py:357-467,
// get_contrastive_denoising_training_group): padded class ids with label noise and noised boxes in logit space, from the batch's
// concatenated targets and the four random tensors the reference draws (same shapes, dtypes and draw order - they stay torch
// calls, so a fixed generator gives the reference's noise).  The reference runs a Python loop per image and per group plus ~40
// element-wise launches on [B, 2 G gmax, 4]-sized tensors; here ONE launch.  Every fp32 operation is the reference's, in its
// order, individually rounded (this file is built with -ffp-contract=off): the result is bit-identical to the torch composition.
#include "common.h"

namespace dfine {

__device__ __forceinline__ float cdn_inverse_sigmoid(float x) {               // arch/utils.py:54-56, eps = 1e-5
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float num = fmaxf(x, 1e-5f), den = fmaxf(1.f - x, 1e-5f);
    return logf(num / den);
}

// one thread per (image, slot): slot j of the 2 G gmax denoising queries = group j / (2 gmax), negative half when
// j % (2 gmax) >= gmax, ground-truth index j % gmax (valid while < the image's target count)
__global__ __launch_bounds__(256) void cdn_group_kernel(const int64_t *__restrict__ labels, const float *__restrict__ boxes,
                                                        const int *__restrict__ offsets, const float *__restrict__ flip_rand,
                                                        const int *__restrict__ rnd_cls, const float *__restrict__ sign01,
                                                        const float *__restrict__ mag, int *__restrict__ cls_out,
                                                        float *__restrict__ box_unact, int bs, int gmax, int total, int num_classes,
                                                        float flip_below, float noise_scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * total) return;
    const int b = i / total, j = i - b * total;
    const int s = j % gmax;
    const bool neg = (j % (2 * gmax)) >= gmax;
    const int o0 = offsets[b], cnt = offsets[b + 1] - o0;
    const bool valid = s < cnt;
    int c = valid ? (int)labels[o0 + s] : num_classes;
    if (valid && flip_rand[i] < flip_below) c = rnd_cls[i];
    cls_out[i] = c;
    float bx[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float4 v = *reinterpret_cast<const float4 *>(boxes + (int64_t)(o0 + s) * 4);
        bx[0] = v.x; bx[1] = v.y; bx[2] = v.z; bx[3] = v.w;
    }
    // box_cxcywh_to_xyxy (half extents of the clamped size), the noise span from the UNclamped size (reference :424-425)
    const float hw = 0.5f * fmaxf(bx[2], 0.f), hh = 0.5f * fmaxf(bx[3], 0.f);
    float xy[4] = {bx[0] - hw, bx[1] - hh, bx[0] + hw, bx[1] + hh};
    const float sw = (bx[2] * 0.5f) * noise_scale, sh = (bx[3] * 0.5f) * noise_scale;
    const float span[4] = {sw, sh, sw, sh};
    const float4 sg = *reinterpret_cast<const float4 *>(sign01 + (int64_t)i * 4), mg = *reinterpret_cast<const float4 *>(mag + (int64_t)i * 4);
    const float sgn[4] = {sg.x, sg.y, sg.z, sg.w}, mgn[4] = {mg.x, mg.y, mg.z, mg.w};
    const float ng = neg ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sign = sgn[k] * 2.0f - 1.0f;
        const float m = (mgn[k] + 1.0f) * ng + mgn[k] * (1.f - ng);
        xy[k] = fminf(fmaxf(xy[k] + sign * m * span[k], 0.f), 1.f);
    }
    float out[4] = {(xy[0] + xy[2]) / 2.f, (xy[1] + xy[3]) / 2.f, xy[2] - xy[0], xy[3] - xy[1]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (out[k] < 0.f) out[k] = -out[k];
        out[k] = cdn_inverse_sigmoid(out[k]);
    }
    *reinterpret_cast<float4 *>(box_unact + (int64_t)i * 4) = make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_cdn_group(const int64_t *labels, const float *boxes, const int *offsets, const float *flip_rand, const int *rnd_cls,
                    const float *sign01, const float *mag, int *cls_out, float *box_unact, int bs, int gmax, int groups,
                    int num_classes, float flip_below, float box_noise_scale, void *stream) {
    if (bs == 0 || gmax == 0 || groups == 0) return DFINE_OK;
    if (!offsets || !flip_rand || !rnd_cls || !sign01 || !mag || !cls_out || !box_unact || bs < 0 || gmax < 0 || groups < 0 ||
        num_classes < 1 || !(flip_below > 0.f) || !(box_noise_scale > 0.f))
        return DFINE_E_BADARG;
    const int64_t n = (int64_t)bs * 2 * groups * gmax;
    if (n >= ((int64_t)1 << 31)) return DFINE_E_BADARG;
    hipLaunchKernelGGL(cdn_group_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, boxes, offsets,
                       flip_rand, rnd_cls, sign01, mag, cls_out, box_unact, bs, gmax, 2 * groups * gmax, num_classes,
                       flip_below, box_noise_scale);
    return check_launch();
}

}  // extern "C"
