// A2 (AIFI) / A6 - multi-head self-attention of the token streams, forward and backward, head_dim = 32.
//
// Reference: nn.MultiheadAttention inside TransformerEncoderLayer (src/d_fine/arch/hybrid_encoder.py:243-290, L = 400
// tokens of the 20x20 map, no mask) and TransformerDecoderLayer.self_attn (src/d_fine/arch/dfine_decoder.py:200,233-255,
// L = 300 + denoising queries <= 500, boolean [L, L] mask, True = blocked).  ATen runs it as in-projection GEMMs +
// scaled_dot_product_attention (AOTriton attn_fwd / bwd_kernel_fuse on ROCm) + out-projection; the projections are the
// linear_act kernels of gemm.hip, this file is softmax(Q K^T / sqrt(d) + mask) V and its gradient.
//
// Shapes are tiny for a 2.5 PFLOP/s chip (B * H * L^2 * d * 4 = 8 GFLOP per layer): the kernels are organised so that
// NOTHING leaves registers between the two GEMMs of each direction, by choosing the MFMA orientation per product:
//   forward / dQ ("S^T orientation", waves split the queries): S^T = K Q^T gives every lane ONE query column and 4 keys per
//     16-key tile, i.e. exactly the key order in which gfx950's LDS transpose-read (ds_read_b64_tr_b16) delivers the V^T / K^T
//     operand of the second product (rows 4g..4g+3 and 16+4g..16+4g+3 of a 32-key step) - the bf16-packed probabilities are
//     the B operand of O^T = V^T P^T as they are.  Row max / sum = in-lane reduction + two cross-lane shuffles (lanes
//     l, l+16, l+32, l+48 share a query).  Keys are streamed in blocks of kKB = 64 through LDS with an online softmax, so L is
//     not limited by the register file.
//   dK / dV ("S orientation", waves split the keys): S = Q K^T puts 4 query rows per 16-query tile in each lane = the
//     q order of the transpose-read of dO^T / Q^T: P and dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS as they are;
//     each wave owns 128 keys (K, V fragments live in registers for the whole kernel) and walks the queries in chunks of 32.
//   The softmax statistics are saved by the forward pass in the exp2 domain (lse2 = m + log2(sum)); delta = rowsum(dO * O) is
//   produced by the dQ kernel, which runs first, and read by the dK/dV kernel.
// Layout: q, k, v, o and the gradients are [B, L, H * 32] views with arbitrary row strides (the packed in-projection output is
// consumed without copies); lse2, delta are [B, H, L] fp32.
#include "common.h"

namespace dfine {

// 2^x of the softmax (re)computation as the bare v_exp_f32 (1 ulp): the arguments are <= 0 (score - row maximum / log-sum-exp) or -inf,
// so exp2f's range checks and denormal rescaling (4 more instructions per element) protect nothing - results below 2^-126 flush to
// 0.  Measured (tools/linear_attn_bench.py, B 32, L 492, masked): forward 68.4 -> 63.4 us, backward 125.8 -> 110.6 us.
__device__ __forceinline__ float attn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

typedef __attribute__((ext_vector_type(8))) __bf16 a_bf16x8;
typedef __attribute__((ext_vector_type(4))) float a_f32x4;
typedef short a_tr4 __attribute__((ext_vector_type(4)));
typedef short a_tr8 __attribute__((ext_vector_type(8)));

constexpr int kAttnThreads = 256;
constexpr int kHD = 32;                 // head dim
constexpr int kKB = 64;                 // keys per LDS block (256 -> 128 -> 64: forward 99.7 -> 68.2 -> 62.8 us at B 32, L 492, masked: less LDS and
                                        // fewer score registers per workgroup = more workgroups per CU to cover each other's load - compute phases)
constexpr int kP40 = 40;                // row pitch (elements) of tiles read with 16-byte fragment loads: conflict-free
constexpr int kP48 = 48;                // row pitch of tiles read with the transpose-read: the 8 rows of a 32-lane half hit disjoint banks
// Head dims up to 64 (DS = 2 slabs of 32 dims; 48 = the AIFI layer of D-FINE-x, src/d_fine/configs.py:182, runs zero-padded to 64):
// rows of 64 elements, pitches 72 (an odd number of 16-byte units) / 80 (40 dwords: rows 0 .. 7 start in banks 0, 40, 16, 56, 32, 8, 48, 24)
template <int DS> struct Pitch { static constexpr int frag = DS == 1 ? kP40 : 72, tr = DS == 1 ? kP48 : 80; };

__device__ __forceinline__ a_bf16x8 ld_frag(const uint16_t *p) { return __builtin_bit_cast(a_bf16x8, *reinterpret_cast<const uint4 *>(p)); }

__device__ __forceinline__ a_bf16x8 tr_frag(const uint16_t *p, int pitch) {
    const a_tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_tr4 __attribute__((address_space(3))) *)p);
    const a_tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_tr4 __attribute__((address_space(3))) *)(p + 16 * pitch));
    return __builtin_bit_cast(a_bf16x8, (a_tr8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }

__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// stage `rows` rows (32 bf16 each, global row stride ld) into an LDS tile with row pitch `pitch`; rows >= valid are zero
template <int DS = 1>
__device__ __forceinline__ void stage_rows(uint16_t *dst, int pitch, const uint16_t *src, int64_t ld, int rows, int valid, int tid) {
    for (int it = tid; it < rows * 4 * DS; it += kAttnThreads) {
        const int r = it / (4 * DS), c = (it % (4 * DS)) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < valid) v = *reinterpret_cast<const uint4 *>(src + (int64_t)r * ld + c);
        *reinterpret_cast<uint4 *>(dst + r * pitch + c) = v;
    }
}

// mask value of 4 consecutive keys for one query: bit e set = blocked
__device__ __forceinline__ uint32_t mask4(const uint8_t *mrow, int key, int L) {
    uint32_t m = 0;
    if (key + 8 <= L) {
        // 4 mask bytes at ANY alignment (rows of L bytes: L = 300 + denoising queries is a multiple of 4 only for every other
        // target count): two aligned words and a funnel shift.  (The byte-wise path below costs 4 loads and 4 branches per
        // 4 keys: forward 61 -> 104 us at L = 498.)  key + 8 <= L keeps the second word inside the row.
        const uintptr_t a = reinterpret_cast<uintptr_t>(mrow + key);
        const uint32_t *wp = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        const uint32_t w = __funnelshift_r(wp[0], wp[1], (uint32_t)(a & 3) * 8u);
        m = ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (key + e < L && mrow[key + e]) m |= 1u << e;
    }
    return m;
}

// ------------------------------------------------------------------------------------------------------------------
// forward: block = (b, h, 64 * QT queries), wave = 16 queries per iteration
template <int QT, int DS = 1>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(const uint16_t *__restrict__ q, const uint16_t *__restrict__ k,
                                                                const uint16_t *__restrict__ v, uint16_t *__restrict__ o,
                                                                float *__restrict__ lse2, const uint8_t *__restrict__ mask,
                                                                const uint8_t *__restrict__ msum,
                                                                int B, int L, int H, int ldq, int ldk, int ldv, int ldo,
                                                                float scale_log2e) {
    constexpr int PF = Pitch<DS>::frag, PT = Pitch<DS>::tr, HD = 32 * DS;
    __shared__ __attribute__((aligned(16))) uint16_t sK[kKB * PF];
    __shared__ __attribute__((aligned(16))) uint16_t sV[kKB * PT];
    const int nk64 = (L + 63) >> 6, nq16 = (L + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int bh = blockIdx.x % (B * H), qblk = blockIdx.x / (B * H);        // blocks of one (b, h) share an XCD (L2 reuse of K, V)
    const int b = bh / H, h = bh - b * H;
    const uint16_t *qb = q + (int64_t)b * L * ldq + h * HD;
    const uint16_t *kb_ = k + (int64_t)b * L * ldk + h * HD;
    const uint16_t *vb = v + (int64_t)b * L * ldv + h * HD;
    const float NEG = -INFINITY;

    int qrow[QT];
    a_bf16x8 qf[QT][DS];
    float m_run[QT], l_run[QT];
    a_f32x4 oacc[QT][2 * DS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qrow[t] = qblk * 64 * QT + t * 64 + wave * 16 + i16;
#pragma unroll
        for (int sl = 0; sl < DS; ++sl) {
            uint4 qv = make_uint4(0, 0, 0, 0);
            if (qrow[t] < L) qv = *reinterpret_cast<const uint4 *>(qb + (int64_t)qrow[t] * ldq + 32 * sl + 8 * g);
            qf[t][sl] = __builtin_bit_cast(a_bf16x8, qv);
        }
        m_run[t] = NEG; l_run[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 2 * DS; ++d) oacc[t][d] = a_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int tr_off = (4 * g + (i16 >> 2)) * PT + 4 * (i16 & 3);

    for (int k0 = 0; k0 < L; k0 += kKB) {
        const int valid = min(kKB, L - k0);
        // mask summary (attn_mask_summary_kernel): a (16 queries x 64 keys) tile is free (0), mixed (1) or blocked (2).  A blocked
        // tile changes nothing (every p is 0): skipped - by the whole workgroup (no staging) when all its query tiles are
        // blocked, else by the wave; a free tile runs without the mask loads.  The denoising mask of the decoder
        // (dfine_decoder.py:200, arch/utils.py:442-455) is block-structured: ~25 % of the tiles are blocked, ~60 % free.
        int code[QT];
        if (msum) {
            bool all_blocked = true;
#pragma unroll
            for (int w = 0; w < 4 * QT; ++w) {
                const int q16 = qblk * 4 * QT + w;
                all_blocked = all_blocked && (q16 >= nq16 || msum[q16 * nk64 + (k0 >> 6)] == 2);
            }
            if (all_blocked) continue;                                   // uniform over the workgroup
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int q16 = qblk * 4 * QT + t * 4 + wave;
                code[t] = q16 >= nq16 ? 2 : msum[q16 * nk64 + (k0 >> 6)];
            }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) code[t] = mask ? 1 : 0;
        }
        __syncthreads();
        stage_rows<DS>(sK, PF, kb_ + (int64_t)k0 * ldk, ldk, kKB, valid, tid);
        stage_rows<DS>(sV, PT, vb + (int64_t)k0 * ldv, ldv, kKB, valid, tid);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (code[t] == 2) continue;                                  // (wave-uniform)
            a_f32x4 s[kKB / 16];
#pragma unroll
            for (int kt = 0; kt < kKB / 16; ++kt) {
                s[kt] = a_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DS; ++sl)
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(sK + (kt * 16 + i16) * PF + 32 * sl + 8 * g), qf[t][sl], s[kt], 0, 0, 0);
            }
            const uint8_t *mrow = code[t] == 1 ? mask + (int64_t)min(qrow[t], L - 1) * L : nullptr;
            float bmax = NEG;
#pragma unroll
            for (int kt = 0; kt < kKB / 16; ++kt) {
                const int key = k0 + kt * 16 + 4 * g;
                const uint32_t mb = mrow ? mask4(mrow, key, L) : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = s[kt][r] * scale_log2e;
                    if (key + r >= L || ((mb >> r) & 1u)) x = NEG;
                    s[kt][r] = x;
                    bmax = fmaxf(bmax, x);
                }
            }
            bmax = xor_max(bmax);
            const float m_new = fmaxf(m_run[t], bmax);
            const float m_use = m_new == NEG ? 0.f : m_new;
            const float alpha = attn_exp2(m_run[t] - m_use);                     // m_run = -inf -> 0
            float psum = 0.f;
            uint32_t pk[kKB / 16][2];
#pragma unroll
            for (int kt = 0; kt < kKB / 16; ++kt) {
                const float p0 = attn_exp2(s[kt][0] - m_use), p1 = attn_exp2(s[kt][1] - m_use);
                const float p2 = attn_exp2(s[kt][2] - m_use), p3 = attn_exp2(s[kt][3] - m_use);
                psum += (p0 + p1) + (p2 + p3);
                pk[kt][0] = pack2(p0, p1); pk[kt][1] = pack2(p2, p3);
            }
            psum = xor_sum(psum);
            l_run[t] = l_run[t] * alpha + psum;
            m_run[t] = m_new;
#pragma unroll
            for (int d = 0; d < 2 * DS; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[t][d][r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < kKB / 32; ++ks) {
                const a_bf16x8 pf = __builtin_bit_cast(a_bf16x8, make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]));
#pragma unroll
                for (int d = 0; d < 2 * DS; ++d)
                    oacc[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV + ks * 32 * PT + tr_off + 16 * d, PT), pf,
                                                                         oacc[t][d], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (qrow[t] >= L) continue;
        const float inv = l_run[t] > 0.f ? 1.f / l_run[t] : 0.f;
        uint16_t *op = o + ((int64_t)b * L + qrow[t]) * ldo + h * HD + 4 * g;
#pragma unroll
        for (int d = 0; d < 2 * DS; ++d) {
            uint2 w;
            w.x = pack2(oacc[t][d][0] * inv, oacc[t][d][1] * inv);
            w.y = pack2(oacc[t][d][2] * inv, oacc[t][d][3] * inv);
            *reinterpret_cast<uint2 *>(op + 16 * d) = w;
        }
        if (g == 0 && lse2) lse2[((int64_t)b * H + h) * L + qrow[t]] = m_run[t] + log2f(l_run[t]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dQ (+ delta): same decomposition as the forward pass; P^T is recomputed from lse2
template <int QT, int DS = 1>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dq_kernel(const uint16_t *__restrict__ q, const uint16_t *__restrict__ k,
                                                                   const uint16_t *__restrict__ v, const uint16_t *__restrict__ o,
                                                                   const uint16_t *__restrict__ dout, const float *__restrict__ lse2,
                                                                   const uint8_t *__restrict__ mask, const uint8_t *__restrict__ msum,
                                                                   uint16_t *__restrict__ dq,
                                                                   float *__restrict__ delta, int B, int L, int H, int ldq, int ldk,
                                                                   int ldv, int ldo, int lddo, int lddq, float scale, float scale_log2e) {
    constexpr int PF = Pitch<DS>::frag, PT = Pitch<DS>::tr, HD = 32 * DS;
    __shared__ __attribute__((aligned(16))) uint16_t sK[kKB * PT];            // 16-byte fragment reads AND transpose-reads
    __shared__ __attribute__((aligned(16))) uint16_t sV[kKB * PF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int bh = blockIdx.x % (B * H), qblk = blockIdx.x / (B * H);
    const int b = bh / H, h = bh - b * H;
    const uint16_t *kb_ = k + (int64_t)b * L * ldk + h * HD;
    const uint16_t *vb = v + (int64_t)b * L * ldv + h * HD;

    int qrow[QT];
    a_bf16x8 qf[QT][DS], dof[QT][DS];
    float lse_q[QT], dl[QT];
    a_f32x4 dacc[QT][2 * DS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qrow[t] = qblk * 64 * QT + t * 64 + wave * 16 + i16;
        lse_q[t] = qrow[t] < L ? lse2[((int64_t)b * H + h) * L + qrow[t]] : 0.f;
        float part = 0.f;
#pragma unroll
        for (int sl = 0; sl < DS; ++sl) {
            uint4 qv = make_uint4(0, 0, 0, 0), dv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
            if (qrow[t] < L) {
                qv = *reinterpret_cast<const uint4 *>(q + ((int64_t)b * L + qrow[t]) * ldq + h * HD + 32 * sl + 8 * g);
                dv = *reinterpret_cast<const uint4 *>(dout + ((int64_t)b * L + qrow[t]) * lddo + h * HD + 32 * sl + 8 * g);
                ov = *reinterpret_cast<const uint4 *>(o + ((int64_t)b * L + qrow[t]) * ldo + h * HD + 32 * sl + 8 * g);
            }
            qf[t][sl] = __builtin_bit_cast(a_bf16x8, qv);
            dof[t][sl] = __builtin_bit_cast(a_bf16x8, dv);
            const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                part += __uint_as_float(dw[jj] << 16) * __uint_as_float(ow[jj] << 16) +
                        __uint_as_float(dw[jj] & 0xffff0000u) * __uint_as_float(ow[jj] & 0xffff0000u);
        }
        dl[t] = xor_sum(part);
        if (g == 0 && qrow[t] < L) delta[((int64_t)b * H + h) * L + qrow[t]] = dl[t];
#pragma unroll
        for (int d = 0; d < 2 * DS; ++d) dacc[t][d] = a_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int tr_off = (4 * g + (i16 >> 2)) * PT + 4 * (i16 & 3);

    const int nk64 = (L + 63) >> 6, nq16 = (L + 15) >> 4;
    for (int k0 = 0; k0 < L; k0 += kKB) {
        const int valid = min(kKB, L - k0);
        int code[QT];                                                    // mask summary of the tile: see attn_fwd_kernel (a blocked tile adds 0 to dQ)
        if (msum) {
            bool all_blocked = true;
#pragma unroll
            for (int w = 0; w < 4 * QT; ++w) {
                const int q16 = qblk * 4 * QT + w;
                all_blocked = all_blocked && (q16 >= nq16 || msum[q16 * nk64 + (k0 >> 6)] == 2);
            }
            if (all_blocked) continue;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int q16 = qblk * 4 * QT + t * 4 + wave;
                code[t] = q16 >= nq16 ? 2 : msum[q16 * nk64 + (k0 >> 6)];
            }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) code[t] = mask ? 1 : 0;
        }
        __syncthreads();
        stage_rows<DS>(sK, PT, kb_ + (int64_t)k0 * ldk, ldk, kKB, valid, tid);
        stage_rows<DS>(sV, PF, vb + (int64_t)k0 * ldv, ldv, kKB, valid, tid);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (code[t] == 2) continue;
            const uint8_t *mrow = code[t] == 1 ? mask + (int64_t)min(qrow[t], L - 1) * L : nullptr;
            uint32_t pk[kKB / 16][2];
#pragma unroll
            for (int kt = 0; kt < kKB / 16; ++kt) {
                a_f32x4 s = a_f32x4{0.f, 0.f, 0.f, 0.f}, dp = a_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DS; ++sl) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(sK + (kt * 16 + i16) * PT + 32 * sl + 8 * g), qf[t][sl], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(sV + (kt * 16 + i16) * PF + 32 * sl + 8 * g), dof[t][sl], dp, 0, 0, 0);
                }
                const int key = k0 + kt * 16 + 4 * g;
                const uint32_t mb = mrow ? mask4(mrow, key, L) : 0u;
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = key + r >= L || ((mb >> r) & 1u);
                    const float p = dead ? 0.f : attn_exp2(s[r] * scale_log2e - lse_q[t]);
                    ds[r] = p * (dp[r] - dl[t]) * scale;
                }
                pk[kt][0] = pack2(ds[0], ds[1]); pk[kt][1] = pack2(ds[2], ds[3]);
                // keep the 16 key tiles in program order: with the one-instruction bf16 packing the scheduler otherwise
                // hoists all fragment reads and MFMAs of the unrolled loop (332 registers, one wave per SIMD, 2x slower)
                if (kt & 1) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < kKB / 32; ++ks) {
                const a_bf16x8 pf = __builtin_bit_cast(a_bf16x8, make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]));
#pragma unroll
                for (int d = 0; d < 2 * DS; ++d)
                    dacc[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK + ks * 32 * PT + tr_off + 16 * d, PT), pf,
                                                                         dacc[t][d], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (qrow[t] >= L) continue;
        uint16_t *op = dq + ((int64_t)b * L + qrow[t]) * lddq + h * HD + 4 * g;
#pragma unroll
        for (int d = 0; d < 2 * DS; ++d) {
            uint2 w;
            w.x = pack2(dacc[t][d][0], dacc[t][d][1]);
            w.y = pack2(dacc[t][d][2], dacc[t][d][3]);
            *reinterpret_cast<uint2 *>(op + 16 * d) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, dK and dV: block = (b, h, 512 keys) = NW waves of 16 KT keys each, queries in chunks of 32 through LDS (double
// buffered).  <8, 4>: 128 keys per wave, 256 threads - ONE wave per SIMD for the decoder's 492 tokens (B H = 256 blocks on 256
// CUs): latency-bound, and slowed down further by whatever shares the chip (143 us alone, 268 us next to the side stream's
// weight-gradient kernels).  <4, 8>: 64 keys per wave, 512 threads, two waves per SIMD and half the accumulators per wave.
// MM: mask form - 0 none, 1 byte mask [L, L], 2 transposed bit mask (compile-time: the unmasked encoder layer pays nothing for it)
template <int KT, int NW, int MM, int DS = 1>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkdv_kernel(const uint16_t *__restrict__ q, const uint16_t *__restrict__ k,
                                                                     const uint16_t *__restrict__ v, const uint16_t *__restrict__ dout,
                                                                     const float *__restrict__ lse2, const float *__restrict__ delta,
                                                                     const uint8_t *__restrict__ mask, const uint32_t *__restrict__ mbits,
                                                                     const uint8_t *__restrict__ msumT /* [64-key group][32-query chunk] */,
                                                                     uint16_t *__restrict__ dk,
                                                                     uint16_t *__restrict__ dv, int B, int L, int H, int ldq, int ldk,
                                                                     int ldv, int lddo, int lddk, int lddv, float scale,
                                                                     float scale_log2e) {
    constexpr int PT = Pitch<DS>::tr, HD = 32 * DS;
    __shared__ __attribute__((aligned(16))) uint16_t sQ[2][32 * PT];
    __shared__ __attribute__((aligned(16))) uint16_t sDO[2][32 * PT];
    __shared__ float sL[2][32], sD[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int bh = blockIdx.x % (B * H), kblk = blockIdx.x / (B * H);
    const int b = bh / H, h = bh - b * H;
    const int kw0 = kblk * (NW * 16 * KT) + wave * 16 * KT;                    // this wave's keys
    const uint16_t *qb = q + (int64_t)b * L * ldq + h * HD;
    const uint16_t *dob = dout + (int64_t)b * L * lddo + h * HD;

    a_bf16x8 kf[KT][DS], vf[KT][DS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int sl = 0; sl < DS; ++sl) {
            const int key = kw0 + kt * 16 + i16;
            uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            if (key < L) {
                kv = *reinterpret_cast<const uint4 *>(k + ((int64_t)b * L + key) * ldk + h * HD + 32 * sl + 8 * g);
                vv = *reinterpret_cast<const uint4 *>(v + ((int64_t)b * L + key) * ldv + h * HD + 32 * sl + 8 * g);
            }
            kf[kt][sl] = __builtin_bit_cast(a_bf16x8, kv);
            vf[kt][sl] = __builtin_bit_cast(a_bf16x8, vv);
        }
    a_f32x4 dka[2 * DS][KT], dva[2 * DS][KT];
#pragma unroll
    for (int d = 0; d < 2 * DS; ++d)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { dka[d][kt] = a_f32x4{0.f, 0.f, 0.f, 0.f}; dva[d][kt] = a_f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int tr_off = (4 * g + (i16 >> 2)) * PT + 4 * (i16 & 3);
    const bool wave_live = kw0 < L;

    // staging of the next 32-query chunk in two halves: the global loads are issued BEFORE the chunk's MFMAs, the LDS writes come
    // after them.  (Load and LDS write in one go made every staging thread sit out the memory latency in front of its compute,
    // once per chunk - with one workgroup per CU nothing else covers it.)
    uint4 st_a = make_uint4(0, 0, 0, 0);
    float st_l = 0.f, st_d = 0.f;
    // mask of a chunk as ONE word per key tile: mbits[key][chunk] holds the 32 queries of the chunk (dfine_attn_mask_bits: the
    // transposed, bit-packed mask).  The byte mask costs 32 single-byte loads per lane and chunk: 98 us masked against 44 us
    // unmasked at the decoder's shape.
    const int W32 = (L + 31) >> 5;
    uint32_t mw[KT], mwn[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { mw[kt] = 0u; mwn[kt] = 0u; }
    int code_next = 1, code = 1;                                          // tile summary of the staged / the current chunk
    auto mask_load = [&](int chunk) {
        if (MM != 2) return;
        if (KT == 4 && msumT && kw0 < L) code_next = msumT[(kw0 >> 6) * ((L + 31) >> 5) + chunk];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = kw0 + kt * 16 + i16;
            mwn[kt] = key < L ? mbits[(int64_t)key * W32 + chunk] : 0u;
        }
    };
    auto stage_load = [&](int q0) {
        const int valid = min(32, L - q0);
        mask_load(q0 >> 5);
        st_a = make_uint4(0, 0, 0, 0);
        if (tid < 128 * DS) {
            const int r = tid / (4 * DS), c = (tid % (4 * DS)) * 8;
            if (r < valid) st_a = *reinterpret_cast<const uint4 *>(qb + (int64_t)(q0 + r) * ldq + c);
        } else if (tid < 256 * DS) {
            const int t2 = tid - 128 * DS, r = t2 / (4 * DS), c = (t2 % (4 * DS)) * 8;
            if (r < valid) st_a = *reinterpret_cast<const uint4 *>(dob + (int64_t)(q0 + r) * lddo + c);
        }
        if (tid < 32) {
            const bool ok = tid < valid;
            st_l = ok ? lse2[((int64_t)b * H + h) * L + q0 + tid] : 0.f;
            st_d = ok ? delta[((int64_t)b * H + h) * L + q0 + tid] : 0.f;
        }
    };
    auto stage_store = [&](int buf) {
        if (tid < 128 * DS) {
            const int r = tid / (4 * DS), c = (tid % (4 * DS)) * 8;
            *reinterpret_cast<uint4 *>(&sQ[buf][r * PT + c]) = st_a;
        } else if (tid < 256 * DS) {
            const int t2 = tid - 128 * DS, r = t2 / (4 * DS), c = (t2 % (4 * DS)) * 8;
            *reinterpret_cast<uint4 *>(&sDO[buf][r * PT + c]) = st_a;
        }
        if (tid < 32) { sL[buf][tid] = st_l; sD[buf][tid] = st_d; }
    };

    const int nchunk = (L + 31) / 32;
    stage_load(0);
    stage_store(0);
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1, q0 = c * 32;
        __syncthreads();                                                    // chunk c staged; every wave is done with chunk c - 1
        const bool more = c + 1 < nchunk;
        if (MM == 2) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) mw[kt] = mwn[kt];
            code = code_next;
        }
        if (more) stage_load(q0 + 32);
        if (!wave_live) { if (more) stage_store(buf ^ 1); continue; }
        if (KT == 4 && msumT && MM == 2) {                                // this wave's 64 keys x the chunk's 32 queries: blocked -> adds nothing
            if (code == 2) { if (more) stage_store(buf ^ 1); continue; }  // (the summary byte came with the chunk's mask words)
            if (code == 0 && MM == 2) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) mw[kt] = 0u;
            }
        }
        const uint16_t *tq = sQ[buf], *tdo = sDO[buf];
        uint32_t pp[KT][2], dsp[KT][2];                                       // bf16 pairs: [key tile][query tile] -> (r0 r1, r2 r3)
        uint32_t pp2[KT][2], dsp2[KT][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            a_bf16x8 qa[DS], da[DS];
#pragma unroll
            for (int sl = 0; sl < DS; ++sl) {
                qa[sl] = ld_frag(tq + (qt * 16 + i16) * PT + 32 * sl + 8 * g);
                da[sl] = ld_frag(tdo + (qt * 16 + i16) * PT + 32 * sl + 8 * g);
            }
            float lq[4], dq_[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { lq[r] = sL[buf][qt * 16 + 4 * g + r]; dq_[r] = sD[buf][qt * 16 + 4 * g + r]; }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                a_f32x4 s = a_f32x4{0.f, 0.f, 0.f, 0.f}, dp = a_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < DS; ++sl) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[sl], kf[kt][sl], s, 0, 0, 0);    // S[q = 4g + r][key = i16]
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[sl], vf[kt][sl], dp, 0, 0, 0);
                }
                const int key = kw0 + kt * 16 + i16;
                float p[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + qt * 16 + 4 * g + r;
                    bool dead = key >= L || qq >= L;
                    if (MM == 2) dead = dead || ((mw[kt] >> (qt * 16 + 4 * g + r)) & 1u);
                    else if (MM == 1 && !dead) dead = mask[(int64_t)qq * L + key] != 0;
                    p[r] = dead ? 0.f : attn_exp2(s[r] * scale_log2e - lq[r]);
                    ds[r] = p[r] * (dp[r] - dq_[r]) * scale;
                }
                if (qt == 0) { pp[kt][0] = pack2(p[0], p[1]); pp[kt][1] = pack2(p[2], p[3]); dsp[kt][0] = pack2(ds[0], ds[1]); dsp[kt][1] = pack2(ds[2], ds[3]); }
                else { pp2[kt][0] = pack2(p[0], p[1]); pp2[kt][1] = pack2(p[2], p[3]); dsp2[kt][0] = pack2(ds[0], ds[1]); dsp2[kt][1] = pack2(ds[2], ds[3]); }
            }
        }
#pragma unroll
        for (int d = 0; d < 2 * DS; ++d) {
            const a_bf16x8 dot = tr_frag(tdo + tr_off + 16 * d, PT);         // dO^T[d][q slots of g]
            const a_bf16x8 qtf = tr_frag(tq + tr_off + 16 * d, PT);          // Q^T
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const a_bf16x8 pb = __builtin_bit_cast(a_bf16x8, make_uint4(pp[kt][0], pp[kt][1], pp2[kt][0], pp2[kt][1]));
                const a_bf16x8 db = __builtin_bit_cast(a_bf16x8, make_uint4(dsp[kt][0], dsp[kt][1], dsp2[kt][0], dsp2[kt][1]));
                dva[d][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pb, dva[d][kt], 0, 0, 0);
                dka[d][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, db, dka[d][kt], 0, 0, 0);
            }
        }
        if (more) stage_store(buf ^ 1);
    }
    if (!wave_live) return;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = kw0 + kt * 16 + i16;
        if (key >= L) continue;
#pragma unroll
        for (int d = 0; d < 2 * DS; ++d) {
            uint2 w;
            w.x = pack2(dka[d][kt][0], dka[d][kt][1]); w.y = pack2(dka[d][kt][2], dka[d][kt][3]);
            *reinterpret_cast<uint2 *>(dk + ((int64_t)b * L + key) * lddk + h * HD + 16 * d + 4 * g) = w;
            w.x = pack2(dva[d][kt][0], dva[d][kt][1]); w.y = pack2(dva[d][kt][2], dva[d][kt][3]);
            *reinterpret_cast<uint2 *>(dv + ((int64_t)b * L + key) * lddv + h * HD + 16 * d + 4 * g) = w;
        }
    }
}

// bits[key][w] bit j = mask[32 w + j][key] != 0 (queries past L: 0)
__global__ void attn_mask_bits_kernel(const uint8_t *__restrict__ mask, uint32_t *__restrict__ bits, int L, int W32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * W32) return;
    const int w = i / L, key = i - w * L;                        // consecutive threads = consecutive keys: coalesced byte reads
    uint32_t word = 0u;
    for (int j = 0; j < 32; ++j) {
        const int qq = 32 * w + j;
        if (qq < L && mask[(int64_t)qq * L + key]) word |= 1u << j;
    }
    bits[(int64_t)key * W32 + w] = word;
}

// Tile summaries of a [L, L] mask: A[q16][k64] for the forward / dQ kernels (16 queries x 64 keys), T[k64][q32] for the dK / dV
// kernel (64 keys x 32 queries): 0 = nothing blocked, 2 = everything blocked (positions past L do not count), 1 = mixed.
__global__ __launch_bounds__(64) void attn_mask_summary_kernel(const uint8_t *__restrict__ mask, uint8_t *__restrict__ sumA,
                                                               uint8_t *__restrict__ sumT, int L) {
    const int nq16 = (L + 15) >> 4, nk64 = (L + 63) >> 6, nq32 = (L + 31) >> 5;
    const int i = blockIdx.x, lane = threadIdx.x;                  // one wave per tile, lane = key column of the tile
    int qa, qb, ka;
    uint8_t *dst;
    if (i < nq16 * nk64) {
        const int q16 = i / nk64, k64 = i - q16 * nk64;
        qa = q16 * 16; qb = qa + 16; ka = k64 * 64; dst = sumA + i;
    } else {
        const int j = i - nq16 * nk64, k64 = j / nq32, q32 = j - k64 * nq32;
        qa = q32 * 32; qb = qa + 32; ka = k64 * 64; dst = sumT + j;
    }
    qb = min(qb, L);
    bool any_blocked = false, any_free = false;
    const int k = ka + lane;
    if (k < L)
        for (int q = qa; q < qb; ++q) {
            const bool bl = mask[(int64_t)q * L + k] != 0;
            any_blocked = any_blocked || bl;
            any_free = any_free || !bl;
        }
    const bool ab = __any(any_blocked), af = __any(any_free);
    if (lane == 0) *dst = af ? (ab ? 1 : 0) : 2;
}

}  // namespace dfine

using namespace dfine;

extern "C" {

static bool attn_args_ok(int B, int L, int H, int hd, const int *lds_, int n) {
    if (B < 1 || L < 1 || H < 1 || (hd != kHD && hd != 2 * kHD)) return false;      // 32, or 64 (two slabs; 48 runs zero-padded to 64)
    for (int i = 0; i < n; ++i) if (lds_[i] < H * hd || (lds_[i] & 7)) return false;     // 16-byte aligned rows
    return true;
}

static size_t mask_summary_t_offset(int L) { return (size_t)(((L + 15) >> 4) * ((L + 63) >> 6) + 15) / 16 * 16; }

int64_t dfine_attn_mask_summary_bytes(int L) { return (int64_t)mask_summary_t_offset(L) + ((L + 63) >> 6) * ((L + 31) >> 5); }

// sum: dfine_attn_mask_summary_bytes(L) bytes - the tile summaries of `mask` for dfine_attn_fwd_ms / dfine_attn_bwd_ms
int dfine_attn_mask_summary(const uint8_t *mask, int L, uint8_t *sum, void *stream) {
    if (L == 0) return DFINE_OK;
    if (!mask || !sum || L < 0) return DFINE_E_BADARG;
    const int n = ((L + 15) >> 4) * ((L + 63) >> 6) + ((L + 63) >> 6) * ((L + 31) >> 5);
    hipLaunchKernelGGL(attn_mask_summary_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, mask, sum,
                       sum + mask_summary_t_offset(L), L);
    return check_launch();
}

int dfine_attn_fwd_ms(const void *q, const void *k, const void *v, void *o, float *lse2, const uint8_t *mask, const uint8_t *mask_summary,
                      int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, float scale, void *stream);

int dfine_attn_fwd(const void *q, const void *k, const void *v, void *o, float *lse2, const uint8_t *mask, int B, int L, int H,
                   int hd, int ldq, int ldk, int ldv, int ldo, float scale, void *stream) {
    return dfine_attn_fwd_ms(q, k, v, o, lse2, mask, nullptr, B, L, H, hd, ldq, ldk, ldv, ldo, scale, stream);
}

// mask_summary (optional, with mask): dfine_attn_mask_summary of the same mask - blocked tiles are skipped, free ones run unmasked
int dfine_attn_fwd_ms(const void *q, const void *k, const void *v, void *o, float *lse2, const uint8_t *mask, const uint8_t *mask_summary,
                      int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, float scale, void *stream) {
    if (B == 0 || L == 0) return DFINE_OK;
    const uint8_t *msum = mask ? mask_summary : nullptr;
    const int lds_[4] = {ldq, ldk, ldv, ldo};
    if (!q || !k || !v || !o || !attn_args_ok(B, L, H, hd, lds_, 4)) return DFINE_E_BADARG;
    const float c = scale * 1.44269504088896340736f;
    const int nq = (L + 63) / 64;
    if (hd == kHD)
        hipLaunchKernelGGL((attn_fwd_kernel<1, 1>), dim3(B * H * nq), dim3(kAttnThreads), 0, (hipStream_t)stream, (const uint16_t *)q,
                           (const uint16_t *)k, (const uint16_t *)v, (uint16_t *)o, lse2, mask, msum, B, L, H, ldq, ldk, ldv, ldo, c);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<1, 2>), dim3(B * H * nq), dim3(kAttnThreads), 0, (hipStream_t)stream, (const uint16_t *)q,
                           (const uint16_t *)k, (const uint16_t *)v, (uint16_t *)o, lse2, mask, msum, B, L, H, ldq, ldk, ldv, ldo, c);
    return check_launch();
}

// mask [L, L] uint8 -> the transposed bit mask of dfine_attn_bwd: bits [L][(L + 31) / 32] uint32, bit j of bits[key][w] = mask[32 w + j][key]
int64_t dfine_attn_mask_bits_words(int L) { return (int64_t)L * ((L + 31) / 32); }

int dfine_attn_mask_bits(const uint8_t *mask, int L, uint32_t *bits, void *stream) {
    if (L == 0) return DFINE_OK;
    if (!mask || !bits || L < 0) return DFINE_E_BADARG;
    const int W32 = (L + 31) / 32, n = L * W32;
    hipLaunchKernelGGL(attn_mask_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mask, bits, L, W32);
    return check_launch();
}

int dfine_attn_bwd_ms(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse2,
                      const uint8_t *mask, const uint32_t *mask_bits, const uint8_t *mask_summary, void *dq, void *dk, void *dv, float *delta,
                      int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale,
                      void *stream);

int dfine_attn_bwd(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse2,
                   const uint8_t *mask, const uint32_t *mask_bits, void *dq, void *dk, void *dv, float *delta, int B, int L, int H,
                   int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale, void *stream) {
    return dfine_attn_bwd_ms(q, k, v, o, dout, lse2, mask, mask_bits, nullptr, dq, dk, dv, delta, B, L, H, hd, ldq, ldk, ldv, ldo, lddo,
                             lddq, lddk, lddv, scale, stream);
}

int dfine_attn_bwd_ms(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse2,
                      const uint8_t *mask, const uint32_t *mask_bits, const uint8_t *mask_summary, void *dq, void *dk, void *dv, float *delta,
                      int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale,
                      void *stream) {
    if (B == 0 || L == 0) return DFINE_OK;
    const uint8_t *msum = mask ? mask_summary : nullptr;
    const uint8_t *msumT = (msum && mask_bits) ? msum + mask_summary_t_offset(L) : nullptr;
    const int lds_[8] = {ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv};
    if (!q || !k || !v || !o || !dout || !lse2 || !dq || !dk || !dv || !delta || !attn_args_ok(B, L, H, hd, lds_, 8))
        return DFINE_E_BADARG;
    const float c = scale * 1.44269504088896340736f;
    hipStream_t st = (hipStream_t)stream;
    const int nq = (L + 63) / 64;
    if (hd == kHD)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<1, 1>), dim3(B * H * nq), dim3(kAttnThreads), 0, st, (const uint16_t *)q, (const uint16_t *)k,
                           (const uint16_t *)v, (const uint16_t *)o, (const uint16_t *)dout, lse2, mask, msum, (uint16_t *)dq, delta, B, L, H,
                           ldq, ldk, ldv, ldo, lddo, lddq, scale, c);
    else
        hipLaunchKernelGGL((attn_bwd_dq_kernel<1, 2>), dim3(B * H * nq), dim3(kAttnThreads), 0, st, (const uint16_t *)q, (const uint16_t *)k,
                           (const uint16_t *)v, (const uint16_t *)o, (const uint16_t *)dout, lse2, mask, msum, (uint16_t *)dq, delta, B, L, H,
                           ldq, ldk, ldv, ldo, lddo, lddq, scale, c);
    if (int e = check_launch()) return e;
    if (hd != kHD) {                                   // two slabs: 8 waves (the staging of a chunk needs 512 threads) of 32 keys each - with 64 keys
        const int nk = (L + 255) / 256;                // per wave the accumulators spill (169 registers)
#define DFINE_DKDV64(MM_) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<2, 8, MM_, 2>), dim3(B * H * nk), dim3(512), 0, st, (const uint16_t *)q, (const uint16_t *)k, \
                             (const uint16_t *)v, (const uint16_t *)dout, lse2, (const float *)delta, mask, mask_bits, msumT, (uint16_t *)dk, (uint16_t *)dv, \
                             B, L, H, ldq, ldk, ldv, lddo, lddk, lddv, scale, c)
        if (!mask) DFINE_DKDV64(0); else if (mask_bits) DFINE_DKDV64(2); else DFINE_DKDV64(1);
#undef DFINE_DKDV64
        return check_launch();
    }
    // keys per workgroup: 512 as 8 waves x 64 keys (measured against <8, 4> (512 keys), <2, 8> (256), <4, 4> (256), <2, 4> (128): fewer keys =
    // more workgroups for the same L, but every workgroup walks all the queries - none was faster at the decoder's L ~ 500)
#define DFINE_DKDV_M(KT_, NW_, MM_)                                                                                                    \
    { const int nk = (L + 16 * KT_ * NW_ - 1) / (16 * KT_ * NW_);                                                                      \
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<KT_, NW_, MM_>), dim3(B * H * nk), dim3(64 * NW_), 0, st, (const uint16_t *)q, (const uint16_t *)k, \
                         (const uint16_t *)v, (const uint16_t *)dout, lse2, (const float *)delta, mask, mask_bits, msumT, (uint16_t *)dk, (uint16_t *)dv, \
                         B, L, H, ldq, ldk, ldv, lddo, lddk, lddv, scale, c); }
#define DFINE_DKDV(KT_, NW_) { if (!mask) DFINE_DKDV_M(KT_, NW_, 0) else if (mask_bits) DFINE_DKDV_M(KT_, NW_, 2) else DFINE_DKDV_M(KT_, NW_, 1) }
    DFINE_DKDV(4, 8)
#undef DFINE_DKDV_M
#undef DFINE_DKDV
    return check_launch();
}

}  // extern "C"
