// A1/A2 - fused BatchNorm2d (+ ReLU/SiLU) (+ learnable scalar affine "LAB"), NCHW, train + eval.
//
// Reference: every ConvBNAct / ConvNormLayer(_fuse) unit runs  bn(conv(x)) -> act -> lab  as separate
// ATen ops (src/d_fine/arch/hgnetv2.py:25-32,75-80; src/d_fine/arch/hybrid_encoder.py:40-45,92-93):
// MIOpen BN stats+apply, an activation kernel and two more elementwise kernels for the scalar affine,
// plus the mirrored chain in backward.  Here (all HBM-bound, fp32 math, bf16 or f32 storage):
//   forward  : bn_stats_kernel      x -> per-(channel, chunk) partial (sum, sum of squares)
//              bn_finalize_kernel   partials -> mean, invstd (float64 combine), running stats
//                                   update (momentum, unbiased var), folded scale/shift
//              bn_apply_kernel      y = lab_s * act(x*scale + shift) + lab_b           (1R + 1W)
//   backward : bn_bwd_reduce_kernel (dy, x) -> partial sums of dz, dz*xhat, dy*act(z), dy
//              bn_bwd_finalize      -> dgamma, dbeta, dlab_s, dlab_b, per-channel coefficients
//              bn_bwd_apply_kernel  dx = scale * (dz - mean(dz) - xhat * mean(dz*xhat))  (2R + 1W)
// with z = x*scale + shift, dz = dy * lab_s * act'(z).  One (channel, image-range) per block; a
// plane (H*W contiguous elements) is read with 16-byte loads when its size allows.
#include <cstdlib>

#include "common.h"

namespace dfine {

constexpr int kBnThreads = 256;
// Workgroups of the flat apply kernels: few fat ones amortise the per-workgroup parameter / finalize prologue on the maps of
// the 640 x 640 models (D-FINE-m: 6.12 -> 5.90 ms per step with 512 instead of 4096; 256: 6.28); the > 64 MB maps of the 960 x 960
// models want the full grid (D-FINE-x + masks: 112.7 -> 109.8 ms per step).
static int bn_grid_cap(int64_t nvec8) {
    return nvec8 <= ((int64_t)1 << 22) ? 512 : 4096;
}
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == ACT_RELU) return fmaxf(z, 0.f);
    // v_rcp_f32 (1 ulp) instead of an IEEE division (~10 instructions): the SiLU passes are VALU-bound, not HBM-bound
    if (act == ACT_SILU) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
    return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == ACT_SILU) {
        const float s = __builtin_amdgcn_rcpf(1.f + __expf(-z));
        return s * (1.f + z * (1.f - s));
    }
    return 1.f;
}

template <int N> __device__ __forceinline__ void block_reduce(float (&v)[N], float *red /*[N][waves]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = kBnThreads / 64;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float x = v[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
        if (lane == 0) red[i * WAVES + wave] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float x = 0.f;
        for (int w = 0; w < WAVES; ++w) x += red[i * WAVES + w];
        v[i] = x;
    }
}

// 8 bf16 (one 16-byte load) -> 8 floats
__device__ __forceinline__ void bf16x8_to_f32(const uint4 &r, float (&o)[8]) {
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
    o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
}

// Index arithmetic without per-vector divisions: a thread walks vectors v, v + step, v + 2*step, ...; the position
// (q, r) = (v / d, v % d) is stepped with (sq, sr) = (step / d, step % d) computed once per kernel.  (An integer
// division is ~25 VALU instructions, a 64-bit one several times that - more than the BN arithmetic of a 16-byte vector.)
struct QR { uint32_t q, r; };
__device__ __forceinline__ QR qr_init(uint32_t v, uint32_t d) { const uint32_t q = v / d; return {q, v - q * d}; }
__device__ __forceinline__ QR qr_step(QR a, uint32_t sq, uint32_t sr, uint32_t d) {
    a.q += sq; a.r += sr;
    if (a.r >= d) { a.r -= d; ++a.q; }
    return a;
}
// channel = plane % C stepped the same way (cq = (step / d) % C); `carry` = the plane advanced by one extra
__device__ __forceinline__ uint32_t chan_step(uint32_t c, uint32_t cq, bool carry, uint32_t C) {
    c += cq + (carry ? 1u : 0u);
    if (c >= C) c -= C;
    if (c >= C) c -= C;
    return c;
}

constexpr int kRedUnroll = 4;      // independent 16-byte loads in flight per thread in the reductions

// partial layout: [C][nchunk][NVAL]
template <typename T>
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const T *__restrict__ x, float *__restrict__ part,
                                                              int C, int HW, int B, int imgs_per_chunk) {
    __shared__ float red[2 * kBnThreads / 64];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int b0 = chunk * imgs_per_chunk, b1 = min(B, b0 + imgs_per_chunk);
    float v[2] = {0.f, 0.f};
    if (sizeof(T) == 2 && (HW & 7) == 0) {
        const int nv = HW >> 3, total = (b1 - b0) * nv;            // (image, 8-vector) pairs of this chunk
        const uint16_t *xb = reinterpret_cast<const uint16_t *>(x);
        const uint32_t sq = kBnThreads / nv, sr = kBnThreads % nv;
        QR pos = qr_init(threadIdx.x, nv);                          // (image, vector) of i
        for (int i0 = threadIdx.x; i0 < total; i0 += kRedUnroll * kBnThreads) {
            uint4 r[kRedUnroll];
#pragma unroll
            for (int j = 0; j < kRedUnroll; ++j) {
                const int i = i0 + j * kBnThreads;
                r[j] = make_uint4(0, 0, 0, 0);
                if (i < total) r[j] = *reinterpret_cast<const uint4 *>(xb + ((int64_t)(b0 + pos.q) * C + c) * HW + pos.r * 8);
                pos = qr_step(pos, sq, sr, nv);
            }
#pragma unroll
            for (int j = 0; j < kRedUnroll; ++j) {
                float a[8];
                bf16x8_to_f32(r[j], a);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[0] += a[e]; v[1] += a[e] * a[e]; }
            }
        }
    } else if ((HW & 3) == 0) {
        const int nv = HW >> 2, total = (b1 - b0) * nv;            // (image, 4-vector) pairs of this chunk
        for (int i = threadIdx.x; i < total; i += kBnThreads) {
            const int bi = i / nv, vi = i - bi * nv;
            const f32x4 a = Vec4<T>::load(x + ((int64_t)(b0 + bi) * C + c) * HW + vi * 4);
            v[0] += (a.x + a.y) + (a.z + a.w);
            v[1] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
    } else {
        for (int b = b0; b < b1; ++b) {
            const T *p = x + ((int64_t)b * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += kBnThreads) {
                const float a = load_f(p + i);
                v[0] += a; v[1] += a * a;
            }
        }
    }
    block_reduce<2>(v, red);
    if (threadIdx.x == 0) {
        float *o = part + ((int64_t)c * gridDim.y + chunk) * 2;
        o[0] = v[0]; o[1] = v[1];
    }
}

// Two tensors of one shape in ONE launch (RepVGG unit: the two convolution outputs of dfine_bn2_act_fwd): blockIdx.z picks the
// tensor - the 40 x 40 / 20 x 20 maps these units run on make a statistics launch ~8 us of launch floor for ~3 us of traffic.
__global__ __launch_bounds__(kBnThreads) void bn_stats2_kernel(const uint16_t *__restrict__ x1, const uint16_t *__restrict__ x2,
                                                               float *__restrict__ part1, float *__restrict__ part2, int C, int HW,
                                                               int B, int imgs_per_chunk) {
    __shared__ float red[2 * kBnThreads / 64];
    const uint16_t *xb = blockIdx.z ? x2 : x1;
    float *part = blockIdx.z ? part2 : part1;
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int b0 = chunk * imgs_per_chunk, b1 = min(B, b0 + imgs_per_chunk);
    float v[2] = {0.f, 0.f};
    const int nv = HW >> 3, total = (b1 - b0) * nv;                // HW % 8 == 0 (bn2_ok)
    const uint32_t sq = kBnThreads / nv, sr = kBnThreads % nv;
    QR pos = qr_init(threadIdx.x, nv);
    for (int i0 = threadIdx.x; i0 < total; i0 += kRedUnroll * kBnThreads) {
        uint4 r[kRedUnroll];
#pragma unroll
        for (int j = 0; j < kRedUnroll; ++j) {
            const int i = i0 + j * kBnThreads;
            r[j] = make_uint4(0, 0, 0, 0);
            if (i < total) r[j] = *reinterpret_cast<const uint4 *>(xb + ((int64_t)(b0 + pos.q) * C + c) * HW + pos.r * 8);
            pos = qr_step(pos, sq, sr, nv);
        }
#pragma unroll
        for (int j = 0; j < kRedUnroll; ++j) {
            float a[8];
            bf16x8_to_f32(r[j], a);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[0] += a[e]; v[1] += a[e] * a[e]; }
        }
    }
    block_reduce<2>(v, red);
    if (threadIdx.x == 0) {
        float *o = part + ((int64_t)c * gridDim.y + chunk) * 2;
        o[0] = v[0]; o[1] = v[1];
    }
}

// one thread per channel
__global__ void bn_finalize_kernel(const float *__restrict__ part, int nchunk, int C, double count,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   float momentum, float eps, float *__restrict__ mean_out,
                                   float *__restrict__ invstd_out, float *__restrict__ scale_out,
                                   float *__restrict__ shift_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        s += (double)part[((int64_t)c * nchunk + k) * 2];
        ss += (double)part[((int64_t)c * nchunk + k) * 2 + 1];
    }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    scale_out[c] = g * invstd;
    shift_out[c] = bt - (float)mean * g * invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// eval / frozen statistics: scale/shift from running stats
__global__ void bn_fold_kernel(int C, const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ running_mean, const float *__restrict__ running_var,
                               float eps, float *__restrict__ scale_out, float *__restrict__ shift_out,
                               float *__restrict__ mean_out, float *__restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    scale_out[c] = g * invstd;
    shift_out[c] = bt - running_mean[c] * g * invstd;
    // what the eval-mode backward takes as save_mean / save_invstd (the host composed them with three ATen launches per unit)
    if (mean_out) mean_out[c] = running_mean[c];
    if (invstd_out) invstd_out[c] = invstd;
}

// Flat variant for planes whose size is a multiple of 4: grid-stride over all 4-vectors of the tensor,
// per-channel constants staged in LDS - keeps every thread busy for small planes (20x20, 40x40), where
// one block per plane leaves most lanes idle.
template <typename T>
__global__ __launch_bounds__(kBnThreads) void bn_apply_flat_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift,
                                                                   const float *__restrict__ lab_s,
                                                                   const float *__restrict__ lab_b, int C, int HW,
                                                                   int64_t nvec, int act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_sc = smem, *s_sh = smem + C;
    for (int i = threadIdx.x; i < C; i += kBnThreads) { s_sc[i] = scale[i]; s_sh[i] = shift[i]; }
    const float ls = lab_s ? lab_s[0] : 1.f, lb = lab_b ? lab_b[0] : 0.f;
    __syncthreads();
    const int nv = HW >> 2;
    for (int64_t v = (int64_t)blockIdx.x * kBnThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kBnThreads) {
        const int c = (int)((v / nv) % C);
        const float sc = s_sc[c], sh = s_sh[c];
        f32x4 a = Vec4<T>::load(x + v * 4);
        a.x = ls * act_fwd(a.x * sc + sh, act) + lb; a.y = ls * act_fwd(a.y * sc + sh, act) + lb;
        a.z = ls * act_fwd(a.z * sc + sh, act) + lb; a.w = ls * act_fwd(a.w * sc + sh, act) + lb;
        Vec4<T>::store(y + v * 4, a);
    }
}


// Finalize folded into the apply kernels (layers with C * nchunk <= kBnFuseMax): part == nullptr -> not fused.
constexpr int kBnFuseMax = 8192;
struct BnFusedFin {
    const float *part;
    int nchunk;
    double count;
    const float *gamma, *beta;
    float *running_mean, *running_var, *mean_out, *invstd_out, *scale_out, *shift_out;
    float momentum, eps;
    const uint16_t *res;               // y += res after the rounding of y (a residual connection behind the unit), or null
};

// the stored bf16 vector plus a residual vector, element-wise in fp32, rounded again: what a separate add of the two bf16 maps gives
__device__ __forceinline__ uint4 bn_add_res(const uint4 &o, const uint16_t *res) {
    const uint4 r = *reinterpret_cast<const uint4 *>(res);
    const uint32_t a[4] = {o.x, o.y, o.z, o.w}, b[4] = {r.x, r.y, r.z, r.w};
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        c[k] = pack_bf16x2(__uint_as_float(a[k] << 16) + __uint_as_float(b[k] << 16),
                           __uint_as_float(a[k] & 0xffff0000u) + __uint_as_float(b[k] & 0xffff0000u));
    return make_uint4(c[0], c[1], c[2], c[3]);
}
struct BnFusedBwdFin {
    const float *part;                 // [C][nchunk][4]
    int nchunk;
    double count;
    float *dgamma, *dbeta, *dlab, *coef;
};

// bf16, HW % 8 == 0: 16-byte vectors, two independent vectors per thread and iteration (nvec < 2^31)
template <int ACT_T>
__global__ __launch_bounds__(kBnThreads) void bn_apply_flat8_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y,
                                                                    const float *__restrict__ scale,
                                                                    const float *__restrict__ shift,
                                                                    const float *__restrict__ lab_s,
                                                                    const float *__restrict__ lab_b, int C, int HW,
                                                                    int64_t nvec, BnFusedFin fin) {
    constexpr int act = ACT_T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_sc = smem, *s_sh = smem + C;
    if (fin.part) {
        // small layers: every block folds the chunk partials of all channels itself (a few KB of L2 reads) instead
        // of waiting for a separate finalize launch; block 0 also publishes the statistics
        for (int c = threadIdx.x; c < C; c += kBnThreads) {
            double sm = 0.0, ss = 0.0;
            for (int k = 0; k < fin.nchunk; ++k) {
                sm += (double)fin.part[((int64_t)c * fin.nchunk + k) * 2];
                ss += (double)fin.part[((int64_t)c * fin.nchunk + k) * 2 + 1];
            }
            const double mean = sm / fin.count;
            double var = ss / fin.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
            const float g = fin.gamma ? fin.gamma[c] : 1.f, bt = fin.beta ? fin.beta[c] : 0.f;
            const float sc = g * invstd, sh = bt - (float)mean * g * invstd;
            s_sc[c] = sc; s_sh[c] = sh;
            if (blockIdx.x == 0) {
                fin.mean_out[c] = (float)mean; fin.invstd_out[c] = invstd;
                fin.scale_out[c] = sc; fin.shift_out[c] = sh;
                if (fin.running_mean) {
                    const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
                    fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
                    fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < C; i += kBnThreads) { s_sc[i] = scale[i]; s_sh[i] = shift[i]; }
    }
    const float ls = lab_s ? lab_s[0] : 1.f, lb = lab_b ? lab_b[0] : 0.f;
    __syncthreads();
    const uint32_t nv = HW >> 3, n = (uint32_t)nvec;
    const uint32_t stride = gridDim.x * kBnThreads;
    const uint32_t sq = stride / nv, sr = stride % nv, cq = sq % C;
    uint32_t v0 = blockIdx.x * kBnThreads + threadIdx.x;
    QR p0 = qr_init(v0, nv);
    uint32_t c0 = p0.q % C;
    for (; v0 < n;) {
        const uint32_t v1 = v0 + stride;
        const QR p1 = qr_step(p0, sq, sr, nv);
        const uint32_t c1 = chan_step(c0, cq, p1.q != p0.q + sq, C);
        const bool has1 = v1 < n;
        const uint4 r0 = *reinterpret_cast<const uint4 *>(x + (int64_t)v0 * 8);
        uint4 r1 = make_uint4(0, 0, 0, 0);
        if (has1) r1 = *reinterpret_cast<const uint4 *>(x + (int64_t)v1 * 8);
        auto apply = [&](const uint4 &r, int64_t v, uint32_t c) {
            const float sc = s_sc[c], sh = s_sh[c];
            float a[8];
            bf16x8_to_f32(r, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = ls * act_fwd(a[e] * sc + sh, act) + lb;
            uint4 o;
            o.x = pack_bf16x2(a[0], a[1]); o.y = pack_bf16x2(a[2], a[3]);
            o.z = pack_bf16x2(a[4], a[5]); o.w = pack_bf16x2(a[6], a[7]);
            if (fin.res) o = bn_add_res(o, fin.res + v * 8);
            *reinterpret_cast<uint4 *>(y + v * 8) = o;
        };
        apply(r0, v0, c0);
        if (has1) apply(r1, v1, c1);
        v0 = v1 + stride;
        p0 = qr_step(p1, sq, sr, nv);
        c0 = chan_step(c1, cq, p0.q != p1.q + sq, C);
    }
}

template <int ACT_T>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply_flat8_kernel(
    const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy, uint16_t *__restrict__ dx,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ lab_s, const float *__restrict__ coef, int C, int HW,
    int64_t nvec, int train, BnFusedBwdFin fin) {
    constexpr int act = ACT_T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_mu = smem, *s_is = smem + C, *s_sc = smem + 2 * C, *s_sh = smem + 3 * C, *s_m0 = smem + 4 * C,
          *s_m1 = smem + 5 * C;
    for (int i = threadIdx.x; i < C; i += kBnThreads) {
        s_mu[i] = mean ? mean[i] : 0.f; s_is[i] = invstd ? invstd[i] : 0.f; s_sc[i] = scale[i]; s_sh[i] = shift[i];
        if (fin.part) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int k = 0; k < fin.nchunk; ++k) {
                const float *p = fin.part + ((int64_t)i * fin.nchunk + k) * 4;
                s0 += p[0]; s1 += p[1]; s2 += p[2]; s3 += p[3];
            }
            const float m0 = (float)(s0 / fin.count), m1 = (float)(s1 / fin.count);
            s_m0[i] = train ? m0 : 0.f; s_m1[i] = train ? m1 : 0.f;
            if (blockIdx.x == 0) {
                if (fin.dgamma) fin.dgamma[i] = (float)s1;
                if (fin.dbeta) fin.dbeta[i] = (float)s0;
                if (fin.dlab) {                                  // zeroed by the caller
                    unsafeAtomicAdd(fin.dlab, (float)s2);
                    unsafeAtomicAdd(fin.dlab + 1, (float)s3);
                }
            }
        } else {
            s_m0[i] = train ? coef[2 * i] : 0.f; s_m1[i] = train ? coef[2 * i + 1] : 0.f;
        }
    }
    const float ls = lab_s ? lab_s[0] : 1.f;
    __syncthreads();
    const uint32_t nv = HW >> 3, n = (uint32_t)nvec;
    const uint32_t stride = gridDim.x * kBnThreads;
    const uint32_t sq = stride / nv, sr = stride % nv, cq = sq % C;
    uint32_t v0 = blockIdx.x * kBnThreads + threadIdx.x;
    QR p0 = qr_init(v0, nv);
    uint32_t c0 = p0.q % C;
    for (; v0 < n;) {
        const uint32_t v1 = v0 + stride;
        const QR p1 = qr_step(p0, sq, sr, nv);
        const uint32_t c1 = chan_step(c0, cq, p1.q != p0.q + sq, C);
        const bool has1 = v1 < n;
        const uint4 x0 = *reinterpret_cast<const uint4 *>(x + (int64_t)v0 * 8), g0 = *reinterpret_cast<const uint4 *>(dy + (int64_t)v0 * 8);
        uint4 x1 = make_uint4(0, 0, 0, 0), g1 = make_uint4(0, 0, 0, 0);
        if (has1) { x1 = *reinterpret_cast<const uint4 *>(x + (int64_t)v1 * 8); g1 = *reinterpret_cast<const uint4 *>(dy + (int64_t)v1 * 8); }
        auto apply = [&](const uint4 &rx, const uint4 &rg, int64_t v, uint32_t c) {
            const float is = s_is[c], sc = s_sc[c], sh = s_sh[c], m1 = s_m1[c];
            // dx = sc * (dz - m0 - xhat * m1), xhat = x * is - mu * is:  dx = dz * sc + (x * k1 + k0)
            const float k1 = -sc * is * m1, k0 = -sc * (s_m0[c] - s_mu[c] * is * m1);
            const float lsc = ls * sc;
            float a[8], g[8], o[8];
            bf16x8_to_f32(rx, a); bf16x8_to_f32(rg, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = a[e] * sc + sh;
                o[e] = (g[e] * lsc) * act_grad(z, act) + (a[e] * k1 + k0);
            }
            uint4 w;
            w.x = pack_bf16x2(o[0], o[1]); w.y = pack_bf16x2(o[2], o[3]);
            w.z = pack_bf16x2(o[4], o[5]); w.w = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<uint4 *>(dx + v * 8) = w;
        };
        apply(x0, g0, v0, c0);
        if (has1) apply(x1, g1, v1, c1);
        v0 = v1 + stride;
        p0 = qr_step(p1, sq, sr, nv);
        c0 = chan_step(c1, cq, p0.q != p1.q + sq, C);
    }
}

template <typename T>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply_flat_kernel(
    const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ dx, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ lab_s, const float *__restrict__ coef, int C, int HW, int64_t nvec, int act,
    int train) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_mu = smem, *s_is = smem + C, *s_sc = smem + 2 * C, *s_sh = smem + 3 * C, *s_m0 = smem + 4 * C,
          *s_m1 = smem + 5 * C;
    for (int i = threadIdx.x; i < C; i += kBnThreads) {
        s_mu[i] = mean ? mean[i] : 0.f; s_is[i] = invstd ? invstd[i] : 0.f; s_sc[i] = scale[i]; s_sh[i] = shift[i];
        s_m0[i] = train ? coef[2 * i] : 0.f; s_m1[i] = train ? coef[2 * i + 1] : 0.f;
    }
    const float ls = lab_s ? lab_s[0] : 1.f;
    __syncthreads();
    const int nv = HW >> 2;
    for (int64_t v = (int64_t)blockIdx.x * kBnThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kBnThreads) {
        const int c = (int)((v / nv) % C);
        const float mu = s_mu[c], is = s_is[c], sc = s_sc[c], sh = s_sh[c], m0 = s_m0[c], m1 = s_m1[c];
        auto f = [&](float xv, float gv) {
            const float z = xv * sc + sh;
            const float dz = gv * ls * act_grad(z, act);
            return sc * (dz - m0 - ((xv - mu) * is) * m1);
        };
        const f32x4 a = Vec4<T>::load(x + v * 4), d = Vec4<T>::load(dy + v * 4);
        Vec4<T>::store(dx + v * 4, {f(a.x, d.x), f(a.y, d.y), f(a.z, d.z), f(a.w, d.w)});
    }
}

// grid: (B*C planes, plane chunks)
template <typename T>
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              const float *__restrict__ lab_s, const float *__restrict__ lab_b,
                                                              int C, int HW, int act) {
    const int plane = blockIdx.x, c = plane % C;
    const float sc = scale[c], sh = shift[c];
    const float ls = lab_s ? lab_s[0] : 1.f, lb = lab_b ? lab_b[0] : 0.f;
    const T *p = x + (int64_t)plane * HW;
    T *q = y + (int64_t)plane * HW;
    const int start = blockIdx.y * kBnThreads * 4 * 4;     // 4 vectors per thread per block
    const int end = min(HW, start + kBnThreads * 16);
    if ((HW & 3) == 0) {
        for (int i = start + threadIdx.x * 4; i < end; i += kBnThreads * 4) {
            f32x4 a = Vec4<T>::load(p + i);
            a.x = ls * act_fwd(a.x * sc + sh, act) + lb;
            a.y = ls * act_fwd(a.y * sc + sh, act) + lb;
            a.z = ls * act_fwd(a.z * sc + sh, act) + lb;
            a.w = ls * act_fwd(a.w * sc + sh, act) + lb;
            Vec4<T>::store(q + i, a);
        }
    } else {
        for (int i = start + threadIdx.x; i < end; i += kBnThreads)
            store_f(q + i, ls * act_fwd(load_f(p + i) * sc + sh, act) + lb);
    }
}

// partials: [C][nchunk][4] = sum dz, sum dz*xhat, sum dy*act(z), sum dy
// ACT_T: activation as a template parameter (no per-element branches); LAB: the two learnable-affine sums are wanted
template <typename T, int ACT_T, bool LAB>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_reduce_kernel(
    const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ lab_s, int C, int HW, int B,
    int imgs_per_chunk) {
    constexpr int act = ACT_T;
    __shared__ float red[4 * kBnThreads / 64];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int b0 = chunk * imgs_per_chunk, b1 = min(B, b0 + imgs_per_chunk);
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    const float ls = LAB ? lab_s[0] : 1.f;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    auto accum = [&](float xv, float g) {
        const float z = xv * sc + sh;
        const float dz = g * ls * act_grad(z, act);
        v[0] += dz;
        v[1] += dz * ((xv - mu) * is);      // not x * is - mu * is: that cancels badly when |mean| >> std (fp32 parity)
        if (LAB) {
            v[2] += g * act_fwd(z, act);
            v[3] += g;
        }
    };
    if (sizeof(T) == 2 && (HW & 7) == 0) {
        const int nv = HW >> 3, total = (b1 - b0) * nv;
        const uint16_t *xb = reinterpret_cast<const uint16_t *>(x), *gb = reinterpret_cast<const uint16_t *>(dy);
        constexpr int U = kRedUnroll;                                 // two streams -> 8 loads in flight
        const uint32_t sq = kBnThreads / nv, sr = kBnThreads % nv;
        QR pos = qr_init(threadIdx.x, nv);                            // (image, vector) of i
        for (int i0 = threadIdx.x; i0 < total; i0 += U * kBnThreads) {
            uint4 rx[U], rg[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int i = i0 + j * kBnThreads;
                rx[j] = make_uint4(0, 0, 0, 0); rg[j] = make_uint4(0, 0, 0, 0);
                if (i < total) {
                    const int64_t off = ((int64_t)(b0 + pos.q) * C + c) * HW + pos.r * 8;
                    rx[j] = *reinterpret_cast<const uint4 *>(xb + off);
                    rg[j] = *reinterpret_cast<const uint4 *>(gb + off);
                }
                pos = qr_step(pos, sq, sr, nv);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float a[8], d[8];
                bf16x8_to_f32(rx[j], a); bf16x8_to_f32(rg[j], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) accum(a[e], d[e]);      // padded lanes carry g = 0 -> contribute 0
            }
        }
    } else if ((HW & 3) == 0) {
        const int nv = HW >> 2, total = (b1 - b0) * nv;
        for (int i = threadIdx.x; i < total; i += kBnThreads) {
            const int bi = i / nv, vi = i - bi * nv;
            const int64_t off = ((int64_t)(b0 + bi) * C + c) * HW + vi * 4;
            const f32x4 a = Vec4<T>::load(x + off), d = Vec4<T>::load(dy + off);
            accum(a.x, d.x); accum(a.y, d.y); accum(a.z, d.z); accum(a.w, d.w);
        }
    } else {
        for (int b = b0; b < b1; ++b) {
            const T *p = x + ((int64_t)b * C + c) * HW;
            const T *g = dy + ((int64_t)b * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += kBnThreads) accum(load_f(p + i), load_f(g + i));
        }
    }
    block_reduce<4>(v, red);
    if (threadIdx.x == 0) {
        float *o = part + ((int64_t)c * gridDim.y + chunk) * 4;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
}

// one thread per channel; lab grads accumulated with atomics into dlab[2] (zeroed by caller)
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ part, int nchunk, int C, double count,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta,
                                       float *__restrict__ dlab, float *__restrict__ coef /*[C][2]*/) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int k = 0; k < nchunk; ++k) {
        const float *p = part + ((int64_t)c * nchunk + k) * 4;
        s0 += p[0]; s1 += p[1]; s2 += p[2]; s3 += p[3];
    }
    if (dgamma) dgamma[c] = (float)s1;
    if (dbeta) dbeta[c] = (float)s0;
    coef[2 * c] = (float)(s0 / count);       // mean(dz)
    coef[2 * c + 1] = (float)(s1 / count);   // mean(dz * xhat)
    if (dlab) {
        unsafeAtomicAdd(dlab, (float)s2);
        unsafeAtomicAdd(dlab + 1, (float)s3);
    }
}

template <typename T>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply_kernel(
    const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ dx, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ lab_s, const float *__restrict__ coef, int C, int HW, int act, int train) {
    const int plane = blockIdx.x, c = plane % C;
    const float mu = mean ? mean[c] : 0.f, is = invstd ? invstd[c] : 0.f, sc = scale[c], sh = shift[c];
    const float ls = lab_s ? lab_s[0] : 1.f;
    const float m0 = train ? coef[2 * c] : 0.f, m1 = train ? coef[2 * c + 1] : 0.f;
    const T *p = x + (int64_t)plane * HW;
    const T *g = dy + (int64_t)plane * HW;
    T *q = dx + (int64_t)plane * HW;
    auto f = [&](float xv, float gv) {
        const float z = xv * sc + sh;
        const float dz = gv * ls * act_grad(z, act);
        return sc * (dz - m0 - ((xv - mu) * is) * m1);
    };
    const int start = blockIdx.y * kBnThreads * 16;
    const int end = min(HW, start + kBnThreads * 16);
    if ((HW & 3) == 0) {
        for (int i = start + threadIdx.x * 4; i < end; i += kBnThreads * 4) {
            const f32x4 a = Vec4<T>::load(p + i), d = Vec4<T>::load(g + i);
            Vec4<T>::store(q + i, {f(a.x, d.x), f(a.y, d.y), f(a.z, d.z), f(a.w, d.w)});
        }
    } else {
        for (int i = start + threadIdx.x; i < end; i += kBnThreads) store_f(q + i, f(load_f(p + i), load_f(g + i)));
    }
}

static int chunks_for(int B, int C, int HW, int *imgs_per_chunk) {
    // One block reduces `per` images of one channel.  Split the batch further only while the grid is still
    // small (< 1024 blocks) AND a block keeps >= 8192 elements of work - thousands of tiny blocks are
    // dominated by the block-reduce / partial-write tail (small planes: 20x20, 40x40).
    int per = B;
    while (per > 1 && (int64_t)C * ((B + per - 1) / per) < 1024 && (int64_t)(per / 2) * HW >= 8192) per = (per + 1) / 2;
    while (per > 1 && (int64_t)per * HW > 262144) per = (per + 1) / 2;
    *imgs_per_chunk = per;
    return (B + per - 1) / per;
}

// ---- RepVGG unit: act(BN_a(x1) + BN_b(x2)) [+ residual] as one op ----------------------------------------------------
// Reference: VGGBlock.forward = act(conv1(x) + conv2(x)) with two ConvNormLayers (src/d_fine/arch/hybrid_encoder.py:106-156)
// and CSPLayer's `bottlenecks(...) + conv2(x)` (hybrid_encoder.py:209-239): BN apply x2, add, activation, add = 11 passes
// over the map in the forward and 13 in the backward (activation backward + two BN backwards).  Here: two statistics
// passes + ONE apply pass (2R [+1R] + 1W), and in the backward ONE reduction pass (3R) + ONE apply pass (3R + 2W), both
// branches sharing dz = dy * act'(z), z = BN_a(x1) + BN_b(x2).  bf16, H*W % 8 == 0, C * nchunk <= kBnFuseMax.
struct Bn2Saved {                      // per-channel rows saved by the forward for the backward
    float *mean, *invstd, *scale, *shift;
};

__device__ __forceinline__ void bn_fold_channel(const BnFusedFin &fin, int c, bool publish, float &sc, float &sh) {
    double sm = 0.0, ss = 0.0;
    for (int k = 0; k < fin.nchunk; ++k) {
        sm += (double)fin.part[((int64_t)c * fin.nchunk + k) * 2];
        ss += (double)fin.part[((int64_t)c * fin.nchunk + k) * 2 + 1];
    }
    const double mean = sm / fin.count;
    double var = ss / fin.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
    const float g = fin.gamma ? fin.gamma[c] : 1.f, bt = fin.beta ? fin.beta[c] : 0.f;
    sc = g * invstd; sh = bt - (float)mean * g * invstd;
    if (publish) {
        fin.mean_out[c] = (float)mean; fin.invstd_out[c] = invstd;
        fin.scale_out[c] = sc; fin.shift_out[c] = sh;
        if (fin.running_mean) {
            const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
            fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
            fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
        }
    }
}

template <int ACT_T, bool RES>
__global__ __launch_bounds__(kBnThreads) void bn2_apply_flat8_kernel(
    const uint16_t *__restrict__ x1, const uint16_t *__restrict__ x2, const uint16_t *__restrict__ res,
    uint16_t *__restrict__ y, int C, int HW, int64_t nvec, BnFusedFin fin1, BnFusedFin fin2) {
    constexpr int act = ACT_T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_a = smem, *s_b = smem + C, *s_sh = smem + 2 * C;
    for (int c = threadIdx.x; c < C; c += kBnThreads) {
        float sa, ha, sb, hb;
        bn_fold_channel(fin1, c, blockIdx.x == 0, sa, ha);
        bn_fold_channel(fin2, c, blockIdx.x == 0, sb, hb);
        s_a[c] = sa; s_b[c] = sb; s_sh[c] = ha + hb;
    }
    __syncthreads();
    const uint32_t nv = HW >> 3, n = (uint32_t)nvec;
    const uint32_t stride = gridDim.x * kBnThreads;
    const uint32_t sq = stride / nv, sr = stride % nv, cq = sq % C;
    uint32_t v = blockIdx.x * kBnThreads + threadIdx.x;
    QR pos = qr_init(v, nv);
    uint32_t c = pos.q % C;
    for (; v < n;) {
        const uint4 r1 = *reinterpret_cast<const uint4 *>(x1 + (int64_t)v * 8);
        const uint4 r2 = *reinterpret_cast<const uint4 *>(x2 + (int64_t)v * 8);
        uint4 rr = make_uint4(0, 0, 0, 0);
        if (RES) rr = *reinterpret_cast<const uint4 *>(res + (int64_t)v * 8);
        const float sa = s_a[c], sb = s_b[c], sh = s_sh[c];
        float a[8], b[8], r[8];
        bf16x8_to_f32(r1, a); bf16x8_to_f32(r2, b); bf16x8_to_f32(rr, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = act_fwd(a[e] * sa + (b[e] * sb + sh), act);
            if (RES) a[e] += r[e];
        }
        uint4 o;
        o.x = pack_bf16x2(a[0], a[1]); o.y = pack_bf16x2(a[2], a[3]);
        o.z = pack_bf16x2(a[4], a[5]); o.w = pack_bf16x2(a[6], a[7]);
        *reinterpret_cast<uint4 *>(y + (int64_t)v * 8) = o;
        v += stride;
        const QR nx = qr_step(pos, sq, sr, nv);
        c = chan_step(c, cq, nx.q != pos.q + sq, C);
        pos = nx;
    }
}

// partials [C][nchunk][4] = sum dz, sum dz * xhat1, sum dz * xhat2, -
template <int ACT_T>
__global__ __launch_bounds__(kBnThreads) void bn2_bwd_reduce_kernel(
    const uint16_t *__restrict__ x1, const uint16_t *__restrict__ x2, const uint16_t *__restrict__ dy,
    float *__restrict__ part, Bn2Saved s1, Bn2Saved s2, int C, int HW, int B, int imgs_per_chunk) {
    constexpr int act = ACT_T;
    __shared__ float red[4 * kBnThreads / 64];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int b0 = chunk * imgs_per_chunk, b1 = min(B, b0 + imgs_per_chunk);
    const float mu1 = s1.mean[c], is1 = s1.invstd[c], sa = s1.scale[c];
    const float mu2 = s2.mean[c], is2 = s2.invstd[c], sb = s2.scale[c];
    const float sh = s1.shift[c] + s2.shift[c];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int nv = HW >> 3, total = (b1 - b0) * nv;
    const uint32_t sq = kBnThreads / nv, sr = kBnThreads % nv;
    QR pos = qr_init(threadIdx.x, nv);
    constexpr int U = 2;                                       // three streams -> 6 loads in flight
    for (int i0 = threadIdx.x; i0 < total; i0 += U * kBnThreads) {
        uint4 ra[U], rb[U], rg[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            ra[j] = make_uint4(0, 0, 0, 0); rb[j] = ra[j]; rg[j] = ra[j];
            if (i0 + j * kBnThreads < total) {
                const int64_t off = ((int64_t)(b0 + pos.q) * C + c) * HW + pos.r * 8;
                ra[j] = *reinterpret_cast<const uint4 *>(x1 + off);
                rb[j] = *reinterpret_cast<const uint4 *>(x2 + off);
                rg[j] = *reinterpret_cast<const uint4 *>(dy + off);
            }
            pos = qr_step(pos, sq, sr, nv);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            float a[8], b[8], g[8];
            bf16x8_to_f32(ra[j], a); bf16x8_to_f32(rb[j], b); bf16x8_to_f32(rg[j], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {                            // padded slots carry g = 0 and contribute nothing
                const float z = a[e] * sa + (b[e] * sb + sh);
                const float dz = g[e] * act_grad(z, act);
                v[0] += dz; v[1] += dz * ((a[e] - mu1) * is1); v[2] += dz * ((b[e] - mu2) * is2);
            }
        }
    }
    block_reduce<4>(v, red);
    if (threadIdx.x == 0) {
        float *o = part + ((int64_t)c * gridDim.y + chunk) * 4;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = 0.f;
    }
}

template <int ACT_T>
__global__ __launch_bounds__(kBnThreads) void bn2_bwd_apply_flat8_kernel(
    const uint16_t *__restrict__ x1, const uint16_t *__restrict__ x2, const uint16_t *__restrict__ dy,
    uint16_t *__restrict__ dx1, uint16_t *__restrict__ dx2, Bn2Saved s1, Bn2Saved s2, const float *__restrict__ part,
    int nchunk, double count, float *__restrict__ dgamma1, float *__restrict__ dbeta1, float *__restrict__ dgamma2,
    float *__restrict__ dbeta2, int C, int HW, int64_t nvec) {
    constexpr int act = ACT_T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // per channel: z = x1 * sa + x2 * sb + sh;  dx1 = dz * sa + (x1 * k1a + k0a);  dx2 = dz * sb + (x2 * k1b + k0b)
    float *s_sa = smem, *s_sb = smem + C, *s_sh = smem + 2 * C, *s_k1a = smem + 3 * C, *s_k0a = smem + 4 * C,
          *s_k1b = smem + 5 * C, *s_k0b = smem + 6 * C;
    for (int c = threadIdx.x; c < C; c += kBnThreads) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int k = 0; k < nchunk; ++k) {
            const float *p = part + ((int64_t)c * nchunk + k) * 4;
            t0 += p[0]; t1 += p[1]; t2 += p[2];
        }
        const float m0 = (float)(t0 / count), m1 = (float)(t1 / count), m2 = (float)(t2 / count);
        const float sa = s1.scale[c], sb = s2.scale[c];
        const float is1 = s1.invstd[c], mu1 = s1.mean[c], is2 = s2.invstd[c], mu2 = s2.mean[c];
        s_sa[c] = sa; s_sb[c] = sb; s_sh[c] = s1.shift[c] + s2.shift[c];
        s_k1a[c] = -sa * is1 * m1; s_k0a[c] = -sa * (m0 - mu1 * is1 * m1);
        s_k1b[c] = -sb * is2 * m2; s_k0b[c] = -sb * (m0 - mu2 * is2 * m2);
        if (blockIdx.x == 0) {
            if (dgamma1) dgamma1[c] = (float)t1;
            if (dbeta1) dbeta1[c] = (float)t0;
            if (dgamma2) dgamma2[c] = (float)t2;
            if (dbeta2) dbeta2[c] = (float)t0;
        }
    }
    __syncthreads();
    const uint32_t nv = HW >> 3, n = (uint32_t)nvec;
    const uint32_t stride = gridDim.x * kBnThreads;
    const uint32_t sq = stride / nv, sr = stride % nv, cq = sq % C;
    uint32_t v = blockIdx.x * kBnThreads + threadIdx.x;
    QR pos = qr_init(v, nv);
    uint32_t c = pos.q % C;
    for (; v < n;) {
        const uint4 r1 = *reinterpret_cast<const uint4 *>(x1 + (int64_t)v * 8);
        const uint4 r2 = *reinterpret_cast<const uint4 *>(x2 + (int64_t)v * 8);
        const uint4 rg = *reinterpret_cast<const uint4 *>(dy + (int64_t)v * 8);
        const float sa = s_sa[c], sb = s_sb[c], sh = s_sh[c];
        const float k1a = s_k1a[c], k0a = s_k0a[c], k1b = s_k1b[c], k0b = s_k0b[c];
        float a[8], b[8], g[8], oa[8], ob[8];
        bf16x8_to_f32(r1, a); bf16x8_to_f32(r2, b); bf16x8_to_f32(rg, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z = a[e] * sa + (b[e] * sb + sh);
            const float dz = g[e] * act_grad(z, act);
            oa[e] = dz * sa + (a[e] * k1a + k0a);
            ob[e] = dz * sb + (b[e] * k1b + k0b);
        }
        uint4 wa, wb;
        wa.x = pack_bf16x2(oa[0], oa[1]); wa.y = pack_bf16x2(oa[2], oa[3]);
        wa.z = pack_bf16x2(oa[4], oa[5]); wa.w = pack_bf16x2(oa[6], oa[7]);
        wb.x = pack_bf16x2(ob[0], ob[1]); wb.y = pack_bf16x2(ob[2], ob[3]);
        wb.z = pack_bf16x2(ob[4], ob[5]); wb.w = pack_bf16x2(ob[6], ob[7]);
        *reinterpret_cast<uint4 *>(dx1 + (int64_t)v * 8) = wa;
        *reinterpret_cast<uint4 *>(dx2 + (int64_t)v * 8) = wb;
        v += stride;
        const QR nx = qr_step(pos, sq, sr, nv);
        c = chan_step(c, cq, nx.q != pos.q + sq, C);
        pos = nx;
    }
}

static bool bn2_ok(int B, int C, int HW, int *nchunk, int *per) {
    if (B < 1 || C < 1 || HW < 8 || (HW & 7) || C > 1024) return false;
    *nchunk = chunks_for(B, C, HW, per);
    return (int64_t)C * *nchunk <= kBnFuseMax && (int64_t)B * C * HW / 8 < (int64_t)1 << 31;
}

// ---- small planes: ONE block per channel holds the channel's B * HW <= 65 536 elements in registers ----------------
// The 40x40 / 20x20 layers (about 100 of the 133 BN units of D-FINE-m) are launch-floor bound with the chunked
// two / three-kernel scheme above (~20 us forward, ~28 us backward per unit for 3-13 MB of data).  Here a 1024-thread
// block reads its channel once (16-byte vectors, <= 8 per thread), reduces through LDS, finalizes the statistics and
// applies / forms dx straight from the registers: one launch, one read pass, no workspace.
constexpr int kBnOneThreads = 1024;
constexpr int kBnOneMaxElems = 65536;

template <int N> __device__ __forceinline__ void block_reduce_one(float (&v)[N], float *red /*[N][16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = kBnOneThreads / 64;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float x = v[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
        if (lane == 0) red[i * WAVES + wave] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) x += red[i * WAVES + w];       // same order in every thread: identical totals
        v[i] = x;
    }
}

template <int VPT, int ACT>
__global__ __launch_bounds__(kBnOneThreads) void bn_one_fwd_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y,
                                                                  const float *__restrict__ lab_s,
                                                                  const float *__restrict__ lab_b, int C, int HW, int B,
                                                                  BnFusedFin fin) {
    constexpr int act = ACT;
    __shared__ float red[2 * kBnOneThreads / 64];
    const int c = blockIdx.x;
    const int nvhw = HW >> 3, nvec = B * nvhw;
    uint4 xv[VPT];
    int64_t off[VPT];
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = threadIdx.x + k * kBnOneThreads;
        const int b = i / nvhw, r = i - b * nvhw;
        off[k] = i < nvec ? ((int64_t)b * C + c) * HW + r * 8 : -1;
        xv[k] = make_uint4(0, 0, 0, 0);
        if (off[k] >= 0) xv[k] = *reinterpret_cast<const uint4 *>(x + off[k]);
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        float a[8];
        bf16x8_to_f32(xv[k], a);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[0] += a[e]; v[1] += a[e] * a[e]; }
    }
    block_reduce_one<2>(v, red);
    const double mean = (double)v[0] / fin.count;
    double var = (double)v[1] / fin.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
    const float g = fin.gamma ? fin.gamma[c] : 1.f, bt = fin.beta ? fin.beta[c] : 0.f;
    const float sc = g * invstd, sh = bt - (float)mean * g * invstd;
    if (threadIdx.x == 0) {
        fin.mean_out[c] = (float)mean; fin.invstd_out[c] = invstd; fin.scale_out[c] = sc; fin.shift_out[c] = sh;
        if (fin.running_mean) {
            const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
            fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
            fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
        }
    }
    const float ls = lab_s ? lab_s[0] : 1.f, lb = lab_b ? lab_b[0] : 0.f;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        if (off[k] < 0) continue;
        float a[8];
        bf16x8_to_f32(xv[k], a);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = ls * act_fwd(a[e] * sc + sh, act) + lb;
        uint4 o;
        o.x = pack_bf16x2(a[0], a[1]); o.y = pack_bf16x2(a[2], a[3]);
        o.z = pack_bf16x2(a[4], a[5]); o.w = pack_bf16x2(a[6], a[7]);
        if (fin.res) o = bn_add_res(o, fin.res + off[k]);
        *reinterpret_cast<uint4 *>(y + off[k]) = o;
    }
}

template <int VPT, int ACT, bool LAB>
__global__ __launch_bounds__(kBnOneThreads) void bn_one_bwd_kernel(
    const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy, uint16_t *__restrict__ dx,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ lab_s, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dlab, int C, int HW, int B, double count) {
    constexpr int act = ACT;
    __shared__ float red[4 * kBnOneThreads / 64];
    const int c = blockIdx.x;
    const int nvhw = HW >> 3, nvec = B * nvhw;
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    const float ls = lab_s ? lab_s[0] : 1.f;
    // element offset of this thread's k-th vector, as 32-bit numbers stepped incrementally (vector i + 1024 is
    // (1024 / nvhw) images and (1024 % nvhw) vectors further): no per-vector divisions, no 64-bit pairs - the
    // 128-VGPR budget of a 1024-thread block is tight with 8 packed vectors resident
    const int qb = kBnOneThreads / nvhw, qr = kBnOneThreads - qb * nvhw;
    int vb0 = threadIdx.x / nvhw, vr0 = threadIdx.x - vb0 * nvhw;
    int offs[VPT];
    {
        int b = vb0, r = vr0;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            offs[k] = (threadIdx.x + k * kBnOneThreads) < nvec ? (b * C + c) * HW + r * 8 : -1;
            r += qr; b += qb;
            if (r >= nvhw) { r -= nvhw; ++b; }
        }
    }
    auto voff = [&](int k) -> int { return offs[k]; };
    // x stays in registers for both passes; dy is kept too only when it fits the 128-VGPR budget (VPT <= 2), otherwise it
    // is read again in the second pass (from L2: the first pass just touched it)
    constexpr bool KEEP_G = VPT <= 2;
    uint4 xv[VPT], gv[KEEP_G ? VPT : 1];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    auto pair_terms = [&](uint32_t xw, uint32_t gw, float &o0, float &o1, bool accumulate, float m0, float m1) {
        const float a0 = __uint_as_float(xw << 16), a1 = __uint_as_float(xw & 0xffff0000u);
        const float g0 = __uint_as_float(gw << 16), g1 = __uint_as_float(gw & 0xffff0000u);
        const float z0 = a0 * sc + sh, z1 = a1 * sc + sh;
        const float d0 = g0 * ls * act_grad(z0, act), d1 = g1 * ls * act_grad(z1, act);
        const float h0 = (a0 - mu) * is, h1 = (a1 - mu) * is;
        if (accumulate) {                                         // padded slots carry g = 0 and contribute nothing
            v[0] += d0 + d1; v[1] += d0 * h0 + d1 * h1;
            if (LAB) { v[2] += g0 * act_fwd(z0, act) + g1 * act_fwd(z1, act); v[3] += g0 + g1; }
        } else {
            o0 = sc * (d0 - m0 - h0 * m1); o1 = sc * (d1 - m0 - h1 * m1);
        }
    };
    float t0, t1;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int o = voff(k);
        uint4 g = make_uint4(0, 0, 0, 0);
        xv[k] = make_uint4(0, 0, 0, 0);
        if (o >= 0) { xv[k] = *reinterpret_cast<const uint4 *>(x + o); g = *reinterpret_cast<const uint4 *>(dy + o); }
        if (KEEP_G) gv[k] = g;
        pair_terms(xv[k].x, g.x, t0, t1, true, 0.f, 0.f); pair_terms(xv[k].y, g.y, t0, t1, true, 0.f, 0.f);
        pair_terms(xv[k].z, g.z, t0, t1, true, 0.f, 0.f); pair_terms(xv[k].w, g.w, t0, t1, true, 0.f, 0.f);
        if (VPT > 2 && (k & 1)) __builtin_amdgcn_sched_barrier(0);   // keep at most two vectors' temporaries live
    }
    block_reduce_one<4>(v, red);
    const float m0 = (float)((double)v[0] / count), m1 = (float)((double)v[1] / count);
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = v[1];
        if (dbeta) dbeta[c] = v[0];
        if (dlab) { unsafeAtomicAdd(dlab, v[2]); unsafeAtomicAdd(dlab + 1, v[3]); }   // zeroed by the caller
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int o = voff(k);
        if (o < 0) continue;
        const uint4 g = KEEP_G ? gv[KEEP_G ? k : 0] : *reinterpret_cast<const uint4 *>(dy + o);
        uint4 w;
        pair_terms(xv[k].x, g.x, t0, t1, false, m0, m1); w.x = pack_bf16x2(t0, t1);
        pair_terms(xv[k].y, g.y, t0, t1, false, m0, m1); w.y = pack_bf16x2(t0, t1);
        pair_terms(xv[k].z, g.z, t0, t1, false, m0, m1); w.z = pack_bf16x2(t0, t1);
        pair_terms(xv[k].w, g.w, t0, t1, false, m0, m1); w.w = pack_bf16x2(t0, t1);
        *reinterpret_cast<uint4 *>(dx + o) = w;
        if (VPT > 2 && (k & 1)) __builtin_amdgcn_sched_barrier(0);
    }
}

// Same one-block-per-channel backward for planes whose x + dy do not fit the registers (B * HW up to 65 536): both
// passes stream the channel from global memory - the second one re-reads what the block itself just pulled into
// L2 - so it is still one launch without workspace or finalize, just not one read pass.
template <int ACT, bool LAB>
__global__ __launch_bounds__(kBnOneThreads) void bn_one_bwd_stream_kernel(
    const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy, uint16_t *__restrict__ dx,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ lab_s, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dlab, int C, int HW, int B, double count) {
    constexpr int act = ACT;
    __shared__ float red[4 * kBnOneThreads / 64];
    const int c = blockIdx.x;
    const int nvhw = HW >> 3, nvec = B * nvhw;
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    const float ls = lab_s ? lab_s[0] : 1.f;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    // both passes walk the channel in batches of kBatch vectors per thread with all loads of a batch in flight (one
    // 1024-thread block per channel has no other latency hiding); positions are stepped, not divided
    constexpr int kBatch = 4;
    const uint32_t sq = kBnOneThreads / nvhw, sr = kBnOneThreads % nvhw;
    auto offset = [&](QR p) -> int64_t { return ((int64_t)p.q * C + c) * HW + p.r * 8; };
    {
        QR pos = qr_init(threadIdx.x, nvhw);
        for (int i0 = threadIdx.x; i0 < nvec; i0 += kBatch * kBnOneThreads) {
            uint4 xr[kBatch], gr[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                xr[k] = make_uint4(0, 0, 0, 0); gr[k] = make_uint4(0, 0, 0, 0);
                if (i0 + k * kBnOneThreads < nvec) {
                    const int64_t o = offset(pos);
                    xr[k] = *reinterpret_cast<const uint4 *>(x + o);
                    gr[k] = *reinterpret_cast<const uint4 *>(dy + o);
                }
                pos = qr_step(pos, sq, sr, nvhw);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                float a[8], g[8];
                bf16x8_to_f32(xr[k], a); bf16x8_to_f32(gr[k], g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {                       // padded slots carry g = 0 and contribute nothing
                    const float z = a[e] * sc + sh;
                    const float dz = g[e] * ls * act_grad(z, act);
                    v[0] += dz; v[1] += dz * ((a[e] - mu) * is);
                    if (LAB) { v[2] += g[e] * act_fwd(z, act); v[3] += g[e]; }
                }
            }
        }
    }
    block_reduce_one<4>(v, red);
    const float m0 = (float)((double)v[0] / count), m1 = (float)((double)v[1] / count);
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = v[1];
        if (dbeta) dbeta[c] = v[0];
        if (dlab) { unsafeAtomicAdd(dlab, v[2]); unsafeAtomicAdd(dlab + 1, v[3]); }   // zeroed by the caller
    }
    const float k1 = -sc * is * m1, k0 = -sc * (m0 - mu * is * m1), lsc = ls * sc;    // dx = dz * sc + (x * k1 + k0)
    {
        QR pos = qr_init(threadIdx.x, nvhw);
        for (int i0 = threadIdx.x; i0 < nvec; i0 += kBatch * kBnOneThreads) {
            uint4 xr[kBatch], gr[kBatch];
            int64_t off[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                off[k] = -1;
                if (i0 + k * kBnOneThreads < nvec) {
                    off[k] = offset(pos);
                    xr[k] = *reinterpret_cast<const uint4 *>(x + off[k]);
                    gr[k] = *reinterpret_cast<const uint4 *>(dy + off[k]);
                }
                pos = qr_step(pos, sq, sr, nvhw);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                if (off[k] < 0) continue;
                float a[8], g[8], w[8];
                bf16x8_to_f32(xr[k], a); bf16x8_to_f32(gr[k], g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = a[e] * sc + sh;
                    w[e] = (g[e] * lsc) * act_grad(z, act) + (a[e] * k1 + k0);
                }
                uint4 ov;
                ov.x = pack_bf16x2(w[0], w[1]); ov.y = pack_bf16x2(w[2], w[3]);
                ov.z = pack_bf16x2(w[4], w[5]); ov.w = pack_bf16x2(w[6], w[7]);
                *reinterpret_cast<uint4 *>(dx + off[k]) = ov;
            }
        }
    }
}

// fp32 maps (BASELINE config #2, no autocast): the same one-block-per-channel scheme with 16-byte vectors of four floats, both
// directions as two streamed passes (the second one re-reads from L2 what the block itself just pulled in): one launch per unit
// and direction instead of three (statistics / finalize / apply: 26 + 39 us per unit on the 40x40 / 20x20 planes of D-FINE-s,
// launch-floor bound).
template <int ACT>
__global__ __launch_bounds__(kBnOneThreads) void bn_one_fwd_f32_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                      const float *__restrict__ lab_s, const float *__restrict__ lab_b,
                                                                      int C, int HW, int B, BnFusedFin fin) {
    constexpr int act = ACT;
    constexpr int kBatch = 4;
    __shared__ float red[2 * kBnOneThreads / 64];
    const int c = blockIdx.x;
    const int nvhw = HW >> 2, nvec = B * nvhw;
    const uint32_t sq = kBnOneThreads / nvhw, sr = kBnOneThreads % nvhw;
    auto offset = [&](QR p) -> int64_t { return ((int64_t)p.q * C + c) * HW + p.r * 4; };
    float v[2] = {0.f, 0.f};
    {
        QR pos = qr_init(threadIdx.x, nvhw);
        for (int i0 = threadIdx.x; i0 < nvec; i0 += kBatch * kBnOneThreads) {
            float4 xr[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + k * kBnOneThreads < nvec) xr[k] = *reinterpret_cast<const float4 *>(x + offset(pos));
                pos = qr_step(pos, sq, sr, nvhw);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                v[0] += (xr[k].x + xr[k].y) + (xr[k].z + xr[k].w);
                v[1] += (xr[k].x * xr[k].x + xr[k].y * xr[k].y) + (xr[k].z * xr[k].z + xr[k].w * xr[k].w);
            }
        }
    }
    block_reduce_one<2>(v, red);
    const double mean = (double)v[0] / fin.count;
    double var = (double)v[1] / fin.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
    const float g = fin.gamma ? fin.gamma[c] : 1.f, bt = fin.beta ? fin.beta[c] : 0.f;
    const float sc = g * invstd, sh = bt - (float)mean * g * invstd;
    if (threadIdx.x == 0) {
        fin.mean_out[c] = (float)mean; fin.invstd_out[c] = invstd; fin.scale_out[c] = sc; fin.shift_out[c] = sh;
        if (fin.running_mean) {
            const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
            fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
            fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
        }
    }
    const float ls = lab_s ? lab_s[0] : 1.f, lb = lab_b ? lab_b[0] : 0.f;
    {
        QR pos = qr_init(threadIdx.x, nvhw);
        for (int i0 = threadIdx.x; i0 < nvec; i0 += kBatch * kBnOneThreads) {
            float4 xr[kBatch];
            int64_t off[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                off[k] = -1;
                if (i0 + k * kBnOneThreads < nvec) { off[k] = offset(pos); xr[k] = *reinterpret_cast<const float4 *>(x + off[k]); }
                pos = qr_step(pos, sq, sr, nvhw);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                if (off[k] < 0) continue;
                float4 o;
                o.x = ls * act_fwd(xr[k].x * sc + sh, act) + lb; o.y = ls * act_fwd(xr[k].y * sc + sh, act) + lb;
                o.z = ls * act_fwd(xr[k].z * sc + sh, act) + lb; o.w = ls * act_fwd(xr[k].w * sc + sh, act) + lb;
                *reinterpret_cast<float4 *>(y + off[k]) = o;
            }
        }
    }
}

template <int ACT, bool LAB>
__global__ __launch_bounds__(kBnOneThreads) void bn_one_bwd_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ lab_s,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dlab, int C, int HW, int B, double count) {
    constexpr int act = ACT;
    constexpr int kBatch = 2;                                           // x and dy: 4 vectors of a thread in flight
    __shared__ float red[4 * kBnOneThreads / 64];
    const int c = blockIdx.x;
    const int nvhw = HW >> 2, nvec = B * nvhw;
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    const float ls = lab_s ? lab_s[0] : 1.f;
    const uint32_t sq = kBnOneThreads / nvhw, sr = kBnOneThreads % nvhw;
    auto offset = [&](QR p) -> int64_t { return ((int64_t)p.q * C + c) * HW + p.r * 4; };
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    {
        QR pos = qr_init(threadIdx.x, nvhw);
        for (int i0 = threadIdx.x; i0 < nvec; i0 += kBatch * kBnOneThreads) {
            float4 xr[kBatch], gr[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                xr[k] = make_float4(0.f, 0.f, 0.f, 0.f); gr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + k * kBnOneThreads < nvec) {
                    const int64_t o = offset(pos);
                    xr[k] = *reinterpret_cast<const float4 *>(x + o);
                    gr[k] = *reinterpret_cast<const float4 *>(dy + o);
                }
                pos = qr_step(pos, sq, sr, nvhw);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const float a[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w}, g[4] = {gr[k].x, gr[k].y, gr[k].z, gr[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {                       // padded slots carry g = 0 and contribute nothing
                    const float z = a[e] * sc + sh;
                    const float dz = g[e] * ls * act_grad(z, act);
                    v[0] += dz; v[1] += dz * ((a[e] - mu) * is);
                    if (LAB) { v[2] += g[e] * act_fwd(z, act); v[3] += g[e]; }
                }
            }
        }
    }
    block_reduce_one<4>(v, red);
    const float m0 = (float)((double)v[0] / count), m1 = (float)((double)v[1] / count);
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = v[1];
        if (dbeta) dbeta[c] = v[0];
        if (dlab) { unsafeAtomicAdd(dlab, v[2]); unsafeAtomicAdd(dlab + 1, v[3]); }   // zeroed by the caller
    }
    {
        QR pos = qr_init(threadIdx.x, nvhw);
        for (int i0 = threadIdx.x; i0 < nvec; i0 += kBatch * kBnOneThreads) {
            float4 xr[kBatch], gr[kBatch];
            int64_t off[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                off[k] = -1;
                if (i0 + k * kBnOneThreads < nvec) {
                    off[k] = offset(pos);
                    xr[k] = *reinterpret_cast<const float4 *>(x + off[k]);
                    gr[k] = *reinterpret_cast<const float4 *>(dy + off[k]);
                }
                pos = qr_step(pos, sq, sr, nvhw);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                if (off[k] < 0) continue;
                const float a[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w}, g[4] = {gr[k].x, gr[k].y, gr[k].z, gr[k].w};
                float w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                       // the expression of bn_bwd_apply_flat_kernel<float>
                    const float z = a[e] * sc + sh;
                    const float dz = g[e] * ls * act_grad(z, act);
                    w[e] = sc * (dz - m0 - ((a[e] - mu) * is) * m1);
                }
                *reinterpret_cast<float4 *>(dx + off[k]) = make_float4(w[0], w[1], w[2], w[3]);
            }
        }
    }
}

static bool bn_one_f32_ok(int dtype, int B, int C, int HW) {
    // (at least ~a quarter of the CUs: a 16-channel unit leaves the chip idle with one block per channel)
    return dtype == DFINE_F32 && (HW & 3) == 0 && (int64_t)B * HW <= kBnOneMaxElems && HW >= 4 && C >= 64;
}

static bool bn_one_ok(int dtype, int B, int HW, int *vpt) {
    if (dtype != DFINE_BF16 || (HW & 7) || (int64_t)B * HW > kBnOneMaxElems) return false;
    const int per = (B * (HW >> 3) + kBnOneThreads - 1) / kBnOneThreads;
    *vpt = per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : 8));
    return true;
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_bn_ws_floats(int B, int C, int HW) {
    int per;
    const int nchunk = chunks_for(B, C, HW, &per);
    return (int64_t)C * nchunk * 4 + 2 * (int64_t)C;   // partials (max of fwd 2 / bwd 4) + coef
}

// One-shot request consumed by the next dfine_bn_act_fwd of the calling thread: y = unit(x) + res (bf16 [B, C, HW], same shape;
// the unit's output is rounded to bf16 first - the sum a separate add pass would give).  HG_Block's residual connection behind the
// aggregation's excitation unit (ref hgnetv2.py:274-275).  Served by the bf16 16-byte-vector kernels (HW % 8 == 0): a launch that
// cannot returns DFINE_E_BADARG and drops the request.
static thread_local const uint16_t *g_bn_res = nullptr;
int dfine_bn_residual_once(const void *res) {
    g_bn_res = (const uint16_t *)res;
    return DFINE_OK;
}

int dfine_bn_act_fwd(const void *x, void *y, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, const float *lab_scale, const float *lab_bias, float *save_mean, float *save_invstd,
                     float *scale, float *shift, float *ws, int dtype, int B, int C, int HW, int act,
                     int training, float momentum, float eps, void *stream) {
    const uint16_t *res = g_bn_res;
    g_bn_res = nullptr;
    if (B == 0 || C == 0 || HW == 0) return DFINE_OK;
    if (!x || !y || !scale || !shift || act < 0 || act > 2) return DFINE_E_BADARG;
    if (dtype != DFINE_F32 && dtype != DFINE_BF16) return DFINE_E_BADARG;
    if (res && (dtype != DFINE_BF16 || (HW & 7) || C > 4096 || (int64_t)B * C * HW / 8 >= (int64_t)1 << 31)) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int cb = (C + 127) / 128;
    bool fuse_fin = false;
    BnFusedFin ffin{};
    ffin.res = res;
    int vpt = 0;
    if (training && save_mean && save_invstd && !res && bn_one_f32_ok(dtype, B, C, HW)) {
        BnFusedFin f1{nullptr, 0, (double)B * HW, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift,
                      momentum, eps, nullptr};
#define DFINE_BN1F32(A) hipLaunchKernelGGL((bn_one_fwd_f32_kernel<A>), dim3(C), dim3(kBnOneThreads), 0, st, (const float *)x, (float *)y, \
                                           lab_scale, lab_bias, C, HW, B, f1)
        if (act == 0) DFINE_BN1F32(0); else if (act == 1) DFINE_BN1F32(1); else DFINE_BN1F32(2);
#undef DFINE_BN1F32
        return check_launch();
    }
    if (training && save_mean && save_invstd && bn_one_ok(dtype, B, HW, &vpt)) {
        BnFusedFin f1{nullptr, 0, (double)B * HW, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift,
                      momentum, eps, res};
#define DFINE_BN1F(V, A) hipLaunchKernelGGL((bn_one_fwd_kernel<V, A>), dim3(C), dim3(kBnOneThreads), 0, st, (const uint16_t *)x, \
                                            (uint16_t *)y, lab_scale, lab_bias, C, HW, B, f1)
#define DFINE_BN1F_A(V) { if (act == 0) DFINE_BN1F(V, 0); else if (act == 1) DFINE_BN1F(V, 1); else DFINE_BN1F(V, 2); }
        if (vpt == 1) DFINE_BN1F_A(1) else if (vpt == 2) DFINE_BN1F_A(2) else if (vpt == 4) DFINE_BN1F_A(4) else DFINE_BN1F_A(8)
#undef DFINE_BN1F_A
#undef DFINE_BN1F
        return check_launch();
    }
    if (training) {
        if (!ws || !save_mean || !save_invstd) return DFINE_E_BADARG;
        int per;
        const int nchunk = chunks_for(B, C, HW, &per);
        if (dtype == DFINE_F32)
            hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(C, nchunk), dim3(kBnThreads), 0, st, (const float *)x, ws, C, HW, B, per);
        else
            hipLaunchKernelGGL(bn_stats_kernel<uint16_t>, dim3(C, nchunk), dim3(kBnThreads), 0, st, (const uint16_t *)x, ws, C, HW, B, per);
        fuse_fin = dtype != DFINE_F32 && (HW & 7) == 0 && C <= 4096 && (int64_t)C * nchunk <= kBnFuseMax &&
                   (int64_t)B * C * HW / 8 < (int64_t)1 << 31;       // = the conditions of the flat8 apply kernel below
        if (fuse_fin)
            ffin = BnFusedFin{ws, nchunk, (double)B * HW, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                              scale, shift, momentum, eps, res};
        else
            hipLaunchKernelGGL(bn_finalize_kernel, dim3(cb), dim3(128), 0, st, ws, nchunk, C, (double)B * HW, gamma, beta,
                               running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
    } else {
        if (!running_mean || !running_var) return DFINE_E_BADARG;
        hipLaunchKernelGGL(bn_fold_kernel, dim3(cb), dim3(128), 0, st, C, gamma, beta, running_mean, running_var, eps, scale, shift,
                           save_mean, save_invstd);
    }
    if ((HW & 3) == 0 && C <= 4096) {
        const int64_t nvec = (int64_t)B * C * HW / 4;
        int64_t nb = (nvec + kBnThreads * 4 - 1) / (kBnThreads * 4);
        if (nb > bn_grid_cap(nvec / 2)) nb = bn_grid_cap(nvec / 2);
        const size_t sm = sizeof(float) * 2 * C;
        if (dtype != DFINE_F32 && (HW & 7) == 0 && nvec / 2 < (int64_t)1 << 31) {
            const int64_t nvec8 = nvec / 2;
            int64_t nb8 = (nvec8 + kBnThreads * 4 - 1) / (kBnThreads * 4);
            if (nb8 > bn_grid_cap(nvec8)) nb8 = bn_grid_cap(nvec8);
#define DFINE_BNA8(A) hipLaunchKernelGGL(bn_apply_flat8_kernel<A>, dim3((unsigned)nb8), dim3(kBnThreads), sm, st, (const uint16_t *)x, \
                                         (uint16_t *)y, scale, shift, lab_scale, lab_bias, C, HW, nvec8, ffin)
            if (act == 0) DFINE_BNA8(0); else if (act == 1) DFINE_BNA8(1); else DFINE_BNA8(2);
#undef DFINE_BNA8
            return check_launch();
        }
        if (dtype == DFINE_F32)
            hipLaunchKernelGGL(bn_apply_flat_kernel<float>, dim3((unsigned)nb), dim3(kBnThreads), sm, st, (const float *)x, (float *)y,
                               scale, shift, lab_scale, lab_bias, C, HW, nvec, act);
        else
            hipLaunchKernelGGL(bn_apply_flat_kernel<uint16_t>, dim3((unsigned)nb), dim3(kBnThreads), sm, st, (const uint16_t *)x,
                               (uint16_t *)y, scale, shift, lab_scale, lab_bias, C, HW, nvec, act);
        return check_launch();
    }
    dim3 grid(B * C, (HW + kBnThreads * 16 - 1) / (kBnThreads * 16));
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(kBnThreads), 0, st, (const float *)x, (float *)y, scale, shift, lab_scale, lab_bias, C, HW, act);
    else
        hipLaunchKernelGGL(bn_apply_kernel<uint16_t>, grid, dim3(kBnThreads), 0, st, (const uint16_t *)x, (uint16_t *)y, scale, shift, lab_scale, lab_bias, C, HW, act);
    return check_launch();
}

int dfine_bn_act_bwd(const void *x, const void *dy, void *dx, const float *save_mean, const float *save_invstd,
                     const float *scale, const float *shift, const float *lab_scale, float *dgamma, float *dbeta,
                     float *dlab, float *ws, int dtype, int B, int C, int HW, int act, int training, void *stream) {
    if (B == 0 || C == 0 || HW == 0) return DFINE_OK;
    if (!x || !dy || !dx || !scale || !shift || !ws || act < 0 || act > 2) return DFINE_E_BADARG;
    if (dtype != DFINE_F32 && dtype != DFINE_BF16) return DFINE_E_BADARG;
    if (training && (!save_mean || !save_invstd)) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int vpt = 0;
    if (training && bn_one_f32_ok(dtype, B, C, HW)) {
#define DFINE_BN1B32(A, LB) hipLaunchKernelGGL((bn_one_bwd_f32_kernel<A, LB>), dim3(C), dim3(kBnOneThreads), 0, st, (const float *)x, \
                                               (const float *)dy, (float *)dx, save_mean, save_invstd, scale, shift, lab_scale, dgamma,  \
                                               dbeta, dlab, C, HW, B, (double)B * HW)
#define DFINE_BN1B32_L(A) { if (lab_scale) DFINE_BN1B32(A, true); else DFINE_BN1B32(A, false); }
        if (act == 0) DFINE_BN1B32_L(0) else if (act == 1) DFINE_BN1B32_L(1) else DFINE_BN1B32_L(2)
#undef DFINE_BN1B32_L
#undef DFINE_BN1B32
        return check_launch();
    }
    // backward: only while x AND dy of the channel fit the 128-VGPR budget of a 1024-thread block (<= 2 vectors per
    // thread = B * HW <= 16 384: the 20x20 planes); the 8-vector instantiation spills ~260 registers
    if (training && bn_one_ok(dtype, B, HW, &vpt) && vpt <= 2) {
#define DFINE_BN1B(V, A, LB) hipLaunchKernelGGL((bn_one_bwd_kernel<V, A, LB>), dim3(C), dim3(kBnOneThreads), 0, st, (const uint16_t *)x, \
                                                (const uint16_t *)dy, (uint16_t *)dx, save_mean, save_invstd, scale, shift, lab_scale, \
                                                dgamma, dbeta, dlab, C, HW, B, (double)B * HW)
#define DFINE_BN1B_L(V, A) { if (lab_scale) DFINE_BN1B(V, A, true); else DFINE_BN1B(V, A, false); }
#define DFINE_BN1B_A(V) { if (act == 0) DFINE_BN1B_L(V, 0) else if (act == 1) DFINE_BN1B_L(V, 1) else DFINE_BN1B_L(V, 2) }
        if (vpt == 1) DFINE_BN1B_A(1) else DFINE_BN1B_A(2)
#undef DFINE_BN1B_L
#undef DFINE_BN1B_A
#undef DFINE_BN1B
        return check_launch();
    }
    // 3 .. 8 vectors per thread: streaming variant (measured equal to the two-kernel path below for C < 256, one launch less)
    if (training && bn_one_ok(dtype, B, HW, &vpt)) {
#define DFINE_BN1S(A, LB) hipLaunchKernelGGL((bn_one_bwd_stream_kernel<A, LB>), dim3(C), dim3(kBnOneThreads), 0, st, (const uint16_t *)x, \
                                             (const uint16_t *)dy, (uint16_t *)dx, save_mean, save_invstd, scale, shift, lab_scale, \
                                             dgamma, dbeta, dlab, C, HW, B, (double)B * HW)
#define DFINE_BN1S_L(A) { if (lab_scale) DFINE_BN1S(A, true); else DFINE_BN1S(A, false); }
        if (act == 0) DFINE_BN1S_L(0) else if (act == 1) DFINE_BN1S_L(1) else DFINE_BN1S_L(2)
#undef DFINE_BN1S_L
#undef DFINE_BN1S
        return check_launch();
    }
    int per;
    const int nchunk = chunks_for(B, C, HW, &per);
    float *coef = ws + (int64_t)C * nchunk * 4;
    // the reductions are needed for the parameter gradients in both modes - and for nothing else in eval mode: a frozen unit
    // (FrozenBatchNorm2d of the D-FINE-l / x backbone: statistics and affine are buffers) skips the pass over x and dy
    const bool need_sums = training || dgamma || dbeta || dlab;
#define DFINE_BNR(TT, A, LB) hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, A, LB>), dim3(C, nchunk), dim3(kBnThreads), 0, st, \
                                                (const TT *)x, (const TT *)dy, ws, save_mean ? save_mean : scale,               \
                                                save_invstd ? save_invstd : scale, scale, shift, lab_scale, C, HW, B, per)
#define DFINE_BNR_A(TT, LB) { if (act == 0) DFINE_BNR(TT, 0, LB); else if (act == 1) DFINE_BNR(TT, 1, LB); else DFINE_BNR(TT, 2, LB); }
    // without a learnable affine the two extra sums stay zero (the partial buffer slots are still written)
    if (!need_sums) {}
    else if (dtype == DFINE_F32) { if (lab_scale) DFINE_BNR_A(float, true) else DFINE_BNR_A(float, false) }
    else { if (lab_scale) DFINE_BNR_A(uint16_t, true) else DFINE_BNR_A(uint16_t, false) }
#undef DFINE_BNR_A
#undef DFINE_BNR
    const bool fuse_fin = need_sums && dtype != DFINE_F32 && (HW & 7) == 0 && (HW & 3) == 0 && C <= 2048 && (int64_t)C * nchunk <= kBnFuseMax &&
                          (int64_t)B * C * HW / 8 < (int64_t)1 << 31;
    BnFusedBwdFin bfin{};
    if (fuse_fin)
        bfin = BnFusedBwdFin{ws, nchunk, (double)B * HW, dgamma, dbeta, dlab, coef};
    else if (need_sums)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, st, ws, nchunk, C, (double)B * HW,
                           dgamma, dbeta, dlab, coef);
    if ((HW & 3) == 0 && C <= 2048) {
        const int64_t nvec = (int64_t)B * C * HW / 4;
        int64_t nb = (nvec + kBnThreads * 4 - 1) / (kBnThreads * 4);
        if (nb > bn_grid_cap(nvec / 2)) nb = bn_grid_cap(nvec / 2);
        const size_t sm = sizeof(float) * 6 * C;
        if (dtype != DFINE_F32 && (HW & 7) == 0 && nvec / 2 < (int64_t)1 << 31) {
            const int64_t nvec8 = nvec / 2;
            int64_t nb8 = (nvec8 + kBnThreads * 4 - 1) / (kBnThreads * 4);
            if (nb8 > bn_grid_cap(nvec8)) nb8 = bn_grid_cap(nvec8);
#define DFINE_BNB8(A) hipLaunchKernelGGL(bn_bwd_apply_flat8_kernel<A>, dim3((unsigned)nb8), dim3(kBnThreads), sm, st, (const uint16_t *)x, \
                                         (const uint16_t *)dy, (uint16_t *)dx, save_mean, save_invstd, scale, shift, lab_scale, coef, \
                                         C, HW, nvec8, training, bfin)
            if (act == 0) DFINE_BNB8(0); else if (act == 1) DFINE_BNB8(1); else DFINE_BNB8(2);
#undef DFINE_BNB8
            return check_launch();
        }
        if (dtype == DFINE_F32)
            hipLaunchKernelGGL(bn_bwd_apply_flat_kernel<float>, dim3((unsigned)nb), dim3(kBnThreads), sm, st, (const float *)x,
                               (const float *)dy, (float *)dx, save_mean, save_invstd, scale, shift, lab_scale, coef, C, HW, nvec,
                               act, training);
        else
            hipLaunchKernelGGL(bn_bwd_apply_flat_kernel<uint16_t>, dim3((unsigned)nb), dim3(kBnThreads), sm, st, (const uint16_t *)x,
                               (const uint16_t *)dy, (uint16_t *)dx, save_mean, save_invstd, scale, shift, lab_scale, coef, C, HW,
                               nvec, act, training);
        return check_launch();
    }
    dim3 grid(B * C, (HW + kBnThreads * 16 - 1) / (kBnThreads * 16));
    if (dtype == DFINE_F32)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(kBnThreads), 0, st, (const float *)x, (const float *)dy, (float *)dx,
                           save_mean, save_invstd, scale, shift, lab_scale, coef, C, HW, act, training);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<uint16_t>, grid, dim3(kBnThreads), 0, st, (const uint16_t *)x, (const uint16_t *)dy,
                           (uint16_t *)dx, save_mean, save_invstd, scale, shift, lab_scale, coef, C, HW, act, training);
    return check_launch();
}

/* scale [C] = gamma / sqrt(running_var + eps), shift [C] = beta - running_mean * scale: an eval-mode BatchNorm as the
 * per-channel affine dfine_conv_affine_once() takes (gamma / beta may be NULL: 1 / 0). */
int dfine_bn_fold(const float *gamma, const float *beta, const float *running_mean, const float *running_var, float eps, int C,
                  float *scale, float *shift, void *stream) {
    if (C == 0) return DFINE_OK;
    if (!running_mean || !running_var || !scale || !shift || C < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, C, gamma, beta, running_mean, running_var,
                       eps, scale, shift, (float *)nullptr, (float *)nullptr);
    return check_launch();
}

/* RepVGG unit  y = act(BN_a(x1) + BN_b(x2)) [+ residual]  (training mode, bf16): see the kernels above. */
int dfine_bn2_supported(int B, int C, int HW) {
    int nchunk, per;
    return bn2_ok(B, C, HW, &nchunk, &per) ? 1 : 0;
}

int64_t dfine_bn2_ws_floats(int B, int C, int HW) {
    int nchunk, per;
    if (!bn2_ok(B, C, HW, &nchunk, &per)) return DFINE_E_BADARG;
    return (int64_t)C * nchunk * 4;
}

int dfine_bn2_act_fwd(const void *x1, const void *x2, const void *residual, void *y, const float *gamma1, const float *beta1,
                      float *running_mean1, float *running_var1, const float *gamma2, const float *beta2,
                      float *running_mean2, float *running_var2, float *saved /* [8][C] */, float *ws, int B, int C, int HW,
                      int act, float momentum1, float eps1, float momentum2, float eps2, void *stream) {
    int nchunk, per;
    if (!x1 || !x2 || !y || !saved || !ws || act < 0 || act > 2 || !bn2_ok(B, C, HW, &nchunk, &per)) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    float *part1 = ws, *part2 = ws + (int64_t)C * nchunk * 2;
    hipLaunchKernelGGL(bn_stats2_kernel, dim3(C, nchunk, 2), dim3(kBnThreads), 0, st, (const uint16_t *)x1, (const uint16_t *)x2, part1, part2,
                       C, HW, B, per);
    const double count = (double)B * HW;
    BnFusedFin f1{part1, nchunk, count, gamma1, beta1, running_mean1, running_var1, saved, saved + C, saved + 2 * C, saved + 3 * C,
                  momentum1, eps1};
    BnFusedFin f2{part2, nchunk, count, gamma2, beta2, running_mean2, running_var2, saved + 4 * C, saved + 5 * C, saved + 6 * C,
                  saved + 7 * C, momentum2, eps2};
    const int64_t nvec8 = (int64_t)B * C * HW / 8;
    int64_t nb = (nvec8 + kBnThreads * 2 - 1) / (kBnThreads * 2);
    if (nb > 1024) nb = 1024;                  // every block folds the partial sums of all channels in its prologue
    const size_t sm = sizeof(float) * 3 * C;
#define DFINE_BN2A(A, R) hipLaunchKernelGGL((bn2_apply_flat8_kernel<A, R>), dim3((unsigned)nb), dim3(kBnThreads), sm, st,       \
                                            (const uint16_t *)x1, (const uint16_t *)x2, (const uint16_t *)residual, (uint16_t *)y, \
                                            C, HW, nvec8, f1, f2)
#define DFINE_BN2A_R(A) { if (residual) DFINE_BN2A(A, true); else DFINE_BN2A(A, false); }
    if (act == 0) DFINE_BN2A_R(0) else if (act == 1) DFINE_BN2A_R(1) else DFINE_BN2A_R(2)
#undef DFINE_BN2A_R
#undef DFINE_BN2A
    return check_launch();
}

int dfine_bn2_act_bwd(const void *x1, const void *x2, const void *dy, void *dx1, void *dx2, const float *saved /* [8][C] */,
                      float *dgamma1, float *dbeta1, float *dgamma2, float *dbeta2, float *ws, int B, int C, int HW, int act,
                      void *stream) {
    int nchunk, per;
    if (!x1 || !x2 || !dy || !dx1 || !dx2 || !saved || !ws || act < 0 || act > 2 || !bn2_ok(B, C, HW, &nchunk, &per))
        return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    float *sv = const_cast<float *>(saved);
    const Bn2Saved s1{sv, sv + C, sv + 2 * C, sv + 3 * C}, s2{sv + 4 * C, sv + 5 * C, sv + 6 * C, sv + 7 * C};
    const int64_t nvec8 = (int64_t)B * C * HW / 8;
    int64_t nb = (nvec8 + kBnThreads * 2 - 1) / (kBnThreads * 2);
    if (nb > 1024) nb = 1024;
    const size_t sm = sizeof(float) * 7 * C;
#define DFINE_BN2B(A)                                                                                                          \
    {                                                                                                                          \
        hipLaunchKernelGGL(bn2_bwd_reduce_kernel<A>, dim3(C, nchunk), dim3(kBnThreads), 0, st, (const uint16_t *)x1,           \
                           (const uint16_t *)x2, (const uint16_t *)dy, ws, s1, s2, C, HW, B, per);                             \
        hipLaunchKernelGGL(bn2_bwd_apply_flat8_kernel<A>, dim3((unsigned)nb), dim3(kBnThreads), sm, st, (const uint16_t *)x1,  \
                           (const uint16_t *)x2, (const uint16_t *)dy, (uint16_t *)dx1, (uint16_t *)dx2, s1, s2, ws, nchunk,   \
                           (double)B * HW, dgamma1, dbeta1, dgamma2, dbeta2, C, HW, nvec8);                                    \
    }
    if (act == 0) DFINE_BN2B(0) else if (act == 1) DFINE_BN2B(1) else DFINE_BN2B(2)
#undef DFINE_BN2B
    return check_launch();
}

}  // extern "C"
