// A5/A6 - residual / gate + LayerNorm of the token streams in one pass each way.
//
// Reference call sites (src/d_fine/arch/dfine_decoder.py): TransformerDecoderLayer.forward
//   target = norm1(target + dropout1(self_attn(...)))                              :238-243
//   target = gateway(target, dropout2(cross))   with Gate.forward
//            norm(sigmoid(g)[:, :D] * x1 + sigmoid(g)[:, D:] * x2), g = gate([x1, x2])  :258-271
//   target = norm3((target + dropout4(ffn(target))).clamp(-65504, 65504))           :250-255
// and the two residual LayerNorms of the encoder's TransformerEncoderLayer (hybrid_encoder.py:243-280).
// Under autocast ATen runs each of these as 4-8 elementwise kernels over the [B*Lq, 256] fp32 stream plus
// dtype casts; here one wave owns a row (64 lanes x D/64 contiguous elements), statistics by wave shuffles:
//   mode 0: z = a + b          mode 1: z = clamp(a + b, -c, c)          mode 2: z = sig(g1) * a + sig(g2) * b
//   y = (z - mean) * rstd * weight + bias      (fp32 out; mean / rstd saved for the backward)
// backward recomputes z from the inputs (no activation copy), forms dz, chains to a / b / g in their storage types
// and accumulates dweight / dbias per lane over the block's rows (one atomic per column and block).
#include "common.h"

namespace dfine {

constexpr int kLnThreads = 256;
constexpr int kLnMaxEpl = 16;             // D <= 1024

__device__ __forceinline__ float ln_load(const void *p, int dt, int64_t i) {
    return dt == DFINE_F32 ? reinterpret_cast<const float *>(p)[i] : bf16_to_f32(reinterpret_cast<const uint16_t *>(p)[i]);
}
__device__ __forceinline__ void ln_store(void *p, int dt, int64_t i, float v) {
    if (dt == DFINE_F32) reinterpret_cast<float *>(p)[i] = v;
    else reinterpret_cast<uint16_t *>(p)[i] = f32_to_bf16(v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// EPL consecutive elements of one lane: one 16-byte (fp32) / 8-byte (bf16) access per 4 elements when EPL % 4 == 0 (the
// dtype branch is per row, wave-uniform), element-wise otherwise
template <int EPL>
__device__ __forceinline__ void ln_load_vec(const void *p, int dt, int64_t i, float (&o)[EPL]) {
    if (EPL % 4 == 0) {
        if (dt == DFINE_F32) {
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + i + e);
                o[e] = v.x; o[e + 1] = v.y; o[e + 2] = v.z; o[e + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p) + i + e);
                o[e] = __uint_as_float(v.x << 16); o[e + 1] = __uint_as_float(v.x & 0xffff0000u);
                o[e + 2] = __uint_as_float(v.y << 16); o[e + 3] = __uint_as_float(v.y & 0xffff0000u);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = ln_load(p, dt, i + e);
    }
}
template <int EPL>
__device__ __forceinline__ void ln_store_vec(void *p, int dt, int64_t i, const float (&v)[EPL]) {
    if (EPL % 4 == 0) {
        if (dt == DFINE_F32) {
#pragma unroll
            for (int e = 0; e < EPL; e += 4)
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + i + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e += 4)
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p) + i + e) =
                    make_uint2(pack_bf16x2(v[e], v[e + 1]), pack_bf16x2(v[e + 2], v[e + 3]));
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) ln_store(p, dt, i + e, v[e]);
    }
}

struct LnArgs {
    const void *a, *b, *gate;
    int a_dt, b_dt, g_dt, mode;
    float clampv;
};

// z of one row for this lane's EPL elements; for mode 2 also the two sigmoids
template <int EPL>
__device__ __forceinline__ void ln_row_z(const LnArgs &p, int64_t row, int D, int lane, float (&z)[EPL], float (&av)[EPL],
                                         float (&bv)[EPL], float (&s1)[EPL], float (&s2)[EPL]) {
    const int64_t base = row * D + lane * EPL;
    ln_load_vec<EPL>(p.a, p.a_dt, base, av);
    if (p.b) ln_load_vec<EPL>(p.b, p.b_dt, base, bv);
    else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) bv[e] = 0.f;
    }
    if (p.mode == 2) {
        ln_load_vec<EPL>(p.gate, p.g_dt, row * 2 * D + lane * EPL, s1);
        ln_load_vec<EPL>(p.gate, p.g_dt, row * 2 * D + D + lane * EPL, s2);
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        if (p.mode == 2) {
            s1[e] = sigmoidf(s1[e]);
            s2[e] = sigmoidf(s2[e]);
            z[e] = s1[e] * av[e] + s2[e] * bv[e];
        } else {
            z[e] = av[e] + bv[e];
            if (p.mode == 1) z[e] = fminf(fmaxf(z[e], -p.clampv), p.clampv);
        }
    }
}

template <int EPL>
__global__ __launch_bounds__(kLnThreads) void ln_fused_fwd_kernel(LnArgs p, const float *__restrict__ weight,
                                                                 const float *__restrict__ bias, float eps,
                                                                 float *__restrict__ y, uint16_t *__restrict__ y16,
                                                                 float *__restrict__ mean_out,
                                                                 float *__restrict__ rstd_out, int64_t rows, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[EPL], bt[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { w[e] = weight[lane * EPL + e]; bt[e] = bias ? bias[lane * EPL + e] : 0.f; }
    for (int64_t row = (int64_t)blockIdx.x * (kLnThreads / 64) + wave; row < rows; row += (int64_t)gridDim.x * (kLnThreads / 64)) {
        float z[EPL], av[EPL], bv[EPL], s1[EPL], s2[EPL];
        ln_row_z<EPL>(p, row, D, lane, z, av, bv, s1, s2);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += z[e];
        const float mean = wave_sum(s) / (float)D;
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const float d = z[e] - mean; v += d * d; }
        const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
        float o[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = (z[e] - mean) * rstd * w[e] + bt[e];
        ln_store_vec<EPL>(y, DFINE_F32, row * D + lane * EPL, o);
        if (y16) ln_store_vec<EPL>(y16, DFINE_BF16, row * D + lane * EPL, o);     // the copy the following GEMMs read
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

template <int EPL>
__global__ __launch_bounds__(kLnThreads) void ln_fused_bwd_kernel(LnArgs p, const float *__restrict__ weight,
                                                                 const float *__restrict__ mean_in,
                                                                 const float *__restrict__ rstd_in, const float *__restrict__ dy,
                                                                 void *__restrict__ da, void *__restrict__ db,
                                                                 void *__restrict__ dgate, float *__restrict__ partial,
                                                                 int64_t rows, int D) {
    __shared__ float red[(kLnThreads / 64) * 64 * 2];           // per wave: [2][64] partial sums of one element slot
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[EPL], gw_acc[EPL], gb_acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { w[e] = weight[lane * EPL + e]; gw_acc[e] = 0.f; gb_acc[e] = 0.f; }
    for (int64_t row = (int64_t)blockIdx.x * (kLnThreads / 64) + wave; row < rows; row += (int64_t)gridDim.x * (kLnThreads / 64)) {
        float z[EPL], av[EPL], bv[EPL], s1[EPL], s2[EPL], g[EPL], xh[EPL];
        ln_row_z<EPL>(p, row, D, lane, z, av, bv, s1, s2);
        const float mean = mean_in[row], rstd = rstd_in[row];
        float c1 = 0.f, c2 = 0.f;
        ln_load_vec<EPL>(dy, DFINE_F32, row * D + lane * EPL, g);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            xh[e] = (z[e] - mean) * rstd;
            gw_acc[e] += g[e] * xh[e];
            gb_acc[e] += g[e];
            const float gyw = g[e] * w[e];
            c1 += gyw; c2 += gyw * xh[e];
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
        float dz[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            dz[e] = rstd * (g[e] * w[e] - c1 - xh[e] * c2);
            if (p.mode == 1) {                                    // clamp passes the gradient strictly inside the range
                const float raw = av[e] + bv[e];
                if (!(raw >= -p.clampv && raw <= p.clampv)) dz[e] = 0.f;
            }
        }
        const int64_t base = row * D + lane * EPL;
        if (p.mode == 2) {
            float t1[EPL], t2[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) { t1[e] = dz[e] * s1[e]; t2[e] = dz[e] * s2[e]; }
            if (da) ln_store_vec<EPL>(da, p.a_dt, base, t1);
            if (db) ln_store_vec<EPL>(db, p.b_dt, base, t2);
            if (dgate) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) { t1[e] *= av[e] * (1.f - s1[e]); t2[e] *= bv[e] * (1.f - s2[e]); }
                ln_store_vec<EPL>(dgate, p.g_dt, row * 2 * D + lane * EPL, t1);
                ln_store_vec<EPL>(dgate, p.g_dt, row * 2 * D + D + lane * EPL, t2);
            }
        } else {
            if (da) ln_store_vec<EPL>(da, p.a_dt, base, dz);
            if (db && p.b) ln_store_vec<EPL>(db, p.b_dt, base, dz);
        }
    }
    // column sums over the block's rows: waves combine through LDS, the block writes ONE partial row; a second tiny
    // kernel adds the partial rows (atomics from ~1000 blocks onto 2 D addresses serialise at the L2: 50 us)
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        red[(wave * 2 + 0) * 64 + lane] = gw_acc[e];
        red[(wave * 2 + 1) * 64 + lane] = gb_acc[e];
        __syncthreads();
        if (wave == 0 && partial) {
            float sw = 0.f, sb = 0.f;
            for (int k = 0; k < kLnThreads / 64; ++k) { sw += red[(k * 2 + 0) * 64 + lane]; sb += red[(k * 2 + 1) * 64 + lane]; }
            partial[((int64_t)blockIdx.x * 2 + 0) * D + lane * EPL + e] = sw;
            partial[((int64_t)blockIdx.x * 2 + 1) * D + lane * EPL + e] = sb;
        }
        __syncthreads();
    }
}

// dweight[c] = sum_blocks partial[b][0][c], dbias[c] = sum_blocks partial[b][1][c]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void ln_colsum_reduce_kernel(const float *__restrict__ partial, int nblocks, int D,
                                                              float *__restrict__ dweight, float *__restrict__ dbias) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;       // 16 columns x 16 split lanes per block
    const int col = blockIdx.x * 16 + c, which = blockIdx.y;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < D) {
        const float *src = partial + (int64_t)which * D + col;
        const int64_t stride = 2 * (int64_t)D;
        int b = q;
        for (; b + 48 < nblocks; b += 64) {                      // four independent loads in flight
            s0 += src[(int64_t)b * stride]; s1 += src[(int64_t)(b + 16) * stride];
            s2 += src[(int64_t)(b + 32) * stride]; s3 += src[(int64_t)(b + 48) * stride];
        }
        for (; b < nblocks; b += 16) s0 += src[(int64_t)b * stride];
    }
    red[q][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && col < D) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][c];
        float *dst = which == 0 ? dweight : dbias;
        if (dst) dst[col] = v;
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

static bool ln_args_ok(int mode, const void *a, const void *b, const void *gate, int a_dt, int b_dt, int g_dt, int D) {
    auto dt_ok = [](int d) { return d == DFINE_F32 || d == DFINE_BF16; };
    if (mode < 0 || mode > 2 || !a || !dt_ok(a_dt) || D < 64 || D % 64 || D / 64 > kLnMaxEpl) return false;
    if (b && !dt_ok(b_dt)) return false;
    if (mode == 2 && (!b || !gate || !dt_ok(g_dt))) return false;
    return true;
}

#define DFINE_LN_DISPATCH(EPLV, CALL)                                   \
    switch (EPLV) {                                                    \
        case 1: { constexpr int E = 1; CALL; break; }                  \
        case 2: { constexpr int E = 2; CALL; break; }                  \
        case 4: { constexpr int E = 4; CALL; break; }                  \
        case 6: { constexpr int E = 6; CALL; break; }                  \
        case 8: { constexpr int E = 8; CALL; break; }                  \
        case 16: { constexpr int E = 16; CALL; break; }                \
        default: return DFINE_E_BADARG;                                \
    }

int dfine_ln_fused_fwd(int mode, const void *a, int a_dt, const void *b, int b_dt, const void *gate, int g_dt,
                       const float *weight, const float *bias, float eps, float clampv, float *y, void *y_bf16,
                       float *mean, float *rstd, int64_t rows, int D, void *stream) {
    if (rows == 0) return DFINE_OK;
    if (!ln_args_ok(mode, a, b, gate, a_dt, b_dt, g_dt, D) || !weight || !y || !mean || !rstd || rows < 0) return DFINE_E_BADARG;
    LnArgs p{a, b, gate, a_dt, b_dt, g_dt, mode, clampv};
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    DFINE_LN_DISPATCH(D / 64, hipLaunchKernelGGL((ln_fused_fwd_kernel<E>), dim3((unsigned)blocks), dim3(kLnThreads), 0,
                                                 (hipStream_t)stream, p, weight, bias, eps, y, (uint16_t *)y_bf16, mean, rstd, rows, D))
    return check_launch();
}

static int ln_bwd_blocks(int64_t rows) {
    int64_t blocks = (rows + 15) / 16;                        // >= 4 rows per wave
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int64_t dfine_ln_fused_bwd_ws_floats(int64_t rows, int D) { return (int64_t)ln_bwd_blocks(rows) * 2 * D; }

// da / db / dgate: storage types of a / b / gate (any may be NULL); dweight / dbias fp32 [D] (overwritten); ws:
// dfine_ln_fused_bwd_ws_floats(rows, D) floats - per-block column partial sums [blocks][2][D] (row 0: weight, row 1: bias), written
// whenever ws is given; dweight / dbias both NULL with ws given = partial sums only (the caller reduces them later, e.g. with
// the step's other deferred reductions: dfine_multi_wgrad_reduce rows {splits = blocks, Cout = D, Cin = taps = CP16 = 1, NP16 = 2 D}).
int dfine_ln_fused_bwd(int mode, const void *a, int a_dt, const void *b, int b_dt, const void *gate, int g_dt,
                       const float *weight, const float *mean, const float *rstd, const float *dy, float clampv,
                       void *da, void *db, void *dgate, float *dweight, float *dbias, float *ws, int64_t rows, int D,
                       void *stream) {
    if (rows == 0) return DFINE_OK;
    if (!ln_args_ok(mode, a, b, gate, a_dt, b_dt, g_dt, D) || !weight || !mean || !rstd || !dy || rows < 0) return DFINE_E_BADARG;
    if ((dweight || dbias) && !ws) return DFINE_E_BADARG;
    const bool affine = ws != nullptr;
    LnArgs p{a, b, gate, a_dt, b_dt, g_dt, mode, clampv};
    const int blocks = ln_bwd_blocks(rows);
    float *partial = affine ? ws : nullptr;
    DFINE_LN_DISPATCH(D / 64, hipLaunchKernelGGL((ln_fused_bwd_kernel<E>), dim3((unsigned)blocks), dim3(kLnThreads), 0,
                                                 (hipStream_t)stream, p, weight, mean, rstd, dy, da, db, dgate, partial,
                                                 rows, D))
    if (int e = check_launch()) return e;
    if (dweight || dbias)
        hipLaunchKernelGGL(ln_colsum_reduce_kernel, dim3((D + 15) / 16, 2), dim3(256), 0, (hipStream_t)stream, ws, blocks, D,
                           dweight, dbias);
    return check_launch();
}

}  // extern "C"
