// BatchNorm sums formed in a convolution's store epilogue (DfineConvEpilogue, include/dfine_hip.h).
//
// Reference: every ConvBNAct / ConvNormLayer runs bn(conv(x)) (src/d_fine/arch/hgnetv2.py:75-80, hybrid_encoder.py:40-45); the
// BatchNorm then reads the map once for its batch statistics before it can normalise it, and - in backward - reads dy and the map
// once for sum(dz) and sum(dz * xhat) before it can form dx.  Both reductions are over values a convolution kernel has just
// produced: the forward convolution holds c = conv(x) in registers (mode 1), the consumer's data-gradient convolution holds
// dy (mode 2: the tile of the BatchNorm input it needs on top is ONE extra read, against the two of the separate reduction
// pass).  A lane of the store loop owns 8 consecutive pixels of one channel: it adds its 8 terms, the lanes of the channel row
// combine with DPP shuffles, and the first lane writes the (workgroup-part, channel) slot of a [nchunk][C][2 | 4] fp32 partial
// buffer - plain stores, every slot written exactly once, summed in a fixed order by the finisher (bnact.hip): deterministic.
#pragma once
#include "common.h"

namespace dfine {

__device__ __forceinline__ float epi_act_fwd(float z, int act) {
    if (act == 1) return fmaxf(z, 0.f);
    if (act == 2) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
    return z;
}
__device__ __forceinline__ float epi_act_grad(float z, int act) {
    if (act == 1) return z > 0.f ? 1.f : 0.f;
    if (act == 2) {
        const float s = __builtin_amdgcn_rcpf(1.f + __expf(-z));
        return s * (1.f + z * (1.f - s));
    }
    return 1.f;
}

__device__ __forceinline__ void epi_unpack8(const uint4 &r, float (&o)[8]) {
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
    o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
}

template <int ACT, bool LAB>
__device__ __forceinline__ void epi_bn_bwd8(const float (&a)[8], const float (&x)[8], float mu, float is, float sc, float sh, float ls,
                                            float (&s)[4]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float z = x[e] * sc + sh;
        const float dz = a[e] * ls * epi_act_grad(z, ACT);
        s[0] += dz;
        s[1] += dz * ((x[e] - mu) * is);      // (not x * is - mu * is: see bn_bwd_reduce_kernel)
        if (LAB) { s[2] += a[e] * epi_act_fwd(z, ACT); s[3] += a[e]; }
    }
}

// the terms of 8 stored outputs `v` of channel n; xv (mode 2): the same 8 elements of the BatchNorm input (same shape as the
// output; loaded by the caller ahead of its stores).  Same arithmetic as bn_stats_kernel / bn_bwd_reduce_kernel (bnact.hip);
// the activation / affine cases are wave-uniform branches around the element loop, not selects inside it.
__device__ __forceinline__ void epi_bn_terms(const DfineConvEpilogue &ep, int n, const uint4 &v, const uint4 &xv, float (&s)[4]) {
    float a[8];
    epi_unpack8(v, a);
    if (ep.mode == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[0] += a[e]; s[1] += a[e] * a[e]; }
        return;
    }
    float x[8];
    epi_unpack8(xv, x);
    const float mu = ep.mean[n], is = ep.invstd[n], sc = ep.scale[n], sh = ep.shift[n];
    if (ep.lab_scale) {
        const float ls = ep.lab_scale[0];
        if (ep.act == 1) epi_bn_bwd8<1, true>(a, x, mu, is, sc, sh, ls, s);
        else if (ep.act == 2) epi_bn_bwd8<2, true>(a, x, mu, is, sc, sh, ls, s);
        else epi_bn_bwd8<0, true>(a, x, mu, is, sc, sh, ls, s);
    } else {
        if (ep.act == 1) epi_bn_bwd8<1, false>(a, x, mu, is, sc, sh, 1.f, s);
        else if (ep.act == 2) epi_bn_bwd8<2, false>(a, x, mu, is, sc, sh, 1.f, s);
        else epi_bn_bwd8<0, false>(a, x, mu, is, sc, sh, 1.f, s);
    }
}

// combine the sums of WIDTH consecutive lanes (a power of two <= 16 that divides the lane's group start) and let the group's
// first lane write slot (n, chunk); `row_ok`: the channel exists
// v + (v of lane ^ 1 | ^ 2 | ^ 4 | ^ 8) as ONE v_add_f32 with a DPP source each (quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror,
// row_mirror: after the two quad steps every lane of a quad holds the quad's sum, so the mirrors pair the right groups).
// __shfl_xor is a ds_bpermute_b32 + wait per step here: ~100 cycles each, 16 of them per row of the store loop.
template <int CTRL> __device__ __forceinline__ float epi_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int WIDTH> __device__ __forceinline__ float epi_row_sum(float v) {
    static_assert(WIDTH == 1 || WIDTH == 2 || WIDTH == 4 || WIDTH == 8 || WIDTH == 16, "lanes of one DPP row");
    if (WIDTH >= 2) v = epi_dpp_add<0xB1>(v);
    if (WIDTH >= 4) v = epi_dpp_add<0x4E>(v);
    if (WIDTH >= 8) v = epi_dpp_add<0x141>(v);
    if (WIDTH >= 16) v = epi_dpp_add<0x140>(v);
    return v;
}

template <int WIDTH>
__device__ __forceinline__ void epi_bn_write(const DfineConvEpilogue &ep, int n, int chunk, float (&s)[4], bool row_ok) {
    const int nv = ep.mode == 1 ? 2 : 4;
    s[0] = epi_row_sum<WIDTH>(s[0]);
    s[1] = epi_row_sum<WIDTH>(s[1]);
    if (ep.mode != 1) { s[2] = epi_row_sum<WIDTH>(s[2]); s[3] = epi_row_sum<WIDTH>(s[3]); }
    if (row_ok && (threadIdx.x & (WIDTH - 1)) == 0) {
        // [chunk][channel][nv]: the rows of one workgroup are consecutive channels = whole cache lines from ONE workgroup
        // ([channel][chunk] put 8 / 16 bytes from every workgroup - and every XCD's L2 - into each line: +16 us on a 58 us layer)
        float *o = ep.part + ((int64_t)chunk * ep.cout + n) * nv;
        if (ep.mode == 1) *reinterpret_cast<float2 *>(o) = make_float2(s[0], s[1]);
        else *reinterpret_cast<float4 *>(o) = make_float4(s[0], s[1], s[2], s[3]);
    }
}

// one-shot request set by dfine_conv_epilogue_once(), consumed by the next convolution launch of the calling thread
bool take_conv_epilogue(DfineConvEpilogue *out);

}  // namespace dfine
