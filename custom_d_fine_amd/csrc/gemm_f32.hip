// fp32 token-stream GEMMs (BASELINE config #2: D-FINE-s without autocast) on the f32-input matrix cores:
//     C[z][M, N] = act(alpha * A[z][M, K] . B[z][N, K]^T + bias[N])        both operands K-contiguous ("NT")
// Replaces, for fp32 tensors, the rocBLAS calls behind nn.Linear (src/d_fine/arch/dfine_decoder.py:33-46,119-178,214-271,
// 828-873, src/d_fine/arch/hybrid_encoder.py:243-290) and the two batched products inside
// F.scaled_dot_product_attention (hybrid_encoder.py:256,277, dfine_decoder.py:200,239).  One kernel serves the forward
// (x . W^T), the data gradient (dY . (W^T)^T), the weight gradient (dY^T . (x^T)^T, reduction over the token rows split into
// chunks -> partial products) and Q K^T / P V of the attention (z = batch x head).
//   block = 256 threads = 4 waves (2 x 2), tile 64 (m) x 64 (n), 16-deep K stages through LDS (rows padded to 20 floats),
//   v_mfma_f32_16x16x4_f32: A lane (row l % 16, k l / 16), B lane (column l % 16, k l / 16), D lane holds rows 4 (l / 16) + i
//   of column l % 16.  Next stage's global loads are issued before the MFMAs of the current one.
#include "common.h"

namespace dfine {

typedef __attribute__((ext_vector_type(4))) float gf_f32x4;
constexpr int kGfThreads = 256, kGfBM = 64, kGfBN = 64, kGfBK = 16, kGfPitch = 20;

__device__ __forceinline__ float gf_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    if (act == 3) return v / (1.f + expf(-v));
    return v;
}

template <bool AKM, bool BKM>   // xKM: the operand is given K-major - A as [K, M] (M contiguous), B as [K, N] (N contiguous): the products
                             // with a transposed first / untransposed second factor (weight gradients, P V, dS^T Q ...) read in place
__global__ __launch_bounds__(kGfThreads) void gemm_f32_nt_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                const float *__restrict__ bias, float *__restrict__ C, int M, int N,
                                                                int K, int lda, int ldb, int ldc, int64_t sa, int64_t sb, int64_t sc,
                                                                int splits, int chunk, float alpha, int act, int nt_n) {
    __shared__ __attribute__((aligned(16))) float sA[2][kGfBM * kGfPitch];               // (K-major operand: [16][68] of it)
    __shared__ __attribute__((aligned(16))) float sB[2][kGfBN * kGfPitch];               // (K-major B: [16][68] of it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, zb = z / splits, zs = z - zb * splits;
    const int tn = blockIdx.x % nt_n, tm = blockIdx.x / nt_n;
    const int m0 = tm * kGfBM, n0 = tn * kGfBN;
    const int kbeg = zs * chunk, kend = min(K, kbeg + chunk);
    const float *Az = A + zb * sa, *Bz = B + zb * sb;
    float *Cz = C + (int64_t)z * sc;
    const bool vec_a = (lda & 3) == 0 && ((size_t)Az & 15) == 0 && (AKM || (kbeg & 3) == 0);
    const bool vec_b = (ldb & 3) == 0 && ((size_t)Bz & 15) == 0 && (BKM || (kbeg & 3) == 0);
    const int lr = tid >> 2, lq = (tid & 3) * 4;                           // staging: row lr, floats lq .. lq + 3 of the stage
    auto load4 = [&](const float *base, int ld, int row, int rmax, int k, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rmax && k < kend) {
            const float *p = base + (int64_t)row * ld + k;
            if (vec && k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);
            else {
                v.x = p[0];
                if (k + 1 < kend) v.y = p[1];
                if (k + 2 < kend) v.z = p[2];
                if (k + 3 < kend) v.w = p[3];
            }
        }
        return v;
    };
    // K-major B: thread -> row k = tid / 16 of the stage, floats 4 (tid % 16) .. + 3 of the 64 tile columns
    constexpr int kPitchK = kGfBN + 4;
    const int kr = tid >> 4, nq = (tid & 15) * 4;
    auto load4k = [&](const float *base, int ld, int c0, int cmax, int k, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n = c0 + nq;
        if (k < kend && n < cmax) {
            const float *p = base + (int64_t)k * ld + n;
            if (vec && n + 3 < cmax) v = *reinterpret_cast<const float4 *>(p);
            else {
                v.x = p[0];
                if (n + 1 < cmax) v.y = p[1];
                if (n + 2 < cmax) v.z = p[2];
                if (n + 3 < cmax) v.w = p[3];
            }
        }
        return v;
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i16 = lane & 15;
    gf_f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = gf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (kend - kbeg + kGfBK - 1) / kGfBK;
    float4 pa = AKM ? load4k(Az, lda, m0, M, kbeg + kr, vec_a) : load4(Az, lda, m0 + lr, M, kbeg + lq, vec_a);
    float4 pb = BKM ? load4k(Bz, ldb, n0, N, kbeg + kr, vec_b) : load4(Bz, ldb, n0 + lr, N, kbeg + lq, vec_b);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (AKM) *reinterpret_cast<float4 *>(&sA[buf][kr * kPitchK + nq]) = pa;
        else *reinterpret_cast<float4 *>(&sA[buf][lr * kGfPitch + lq]) = pa;
        if (BKM) *reinterpret_cast<float4 *>(&sB[buf][kr * kPitchK + nq]) = pb;
        else *reinterpret_cast<float4 *>(&sB[buf][lr * kGfPitch + lq]) = pb;
        __syncthreads();
        if (kt + 1 < nk) {
            const int k = kbeg + (kt + 1) * kGfBK + lq;
            const int kk_ = kbeg + (kt + 1) * kGfBK + kr;
            pa = AKM ? load4k(Az, lda, m0, M, kk_, vec_a) : load4(Az, lda, m0 + lr, M, k, vec_a);
            pb = BKM ? load4k(Bz, ldb, n0, N, kk_, vec_b) : load4(Bz, ldb, n0 + lr, N, k, vec_b);
        }
        const float *la = &sA[buf][(wm * 32 + i16) * kGfPitch + g], *lb = &sB[buf][(wn * 32 + i16) * kGfPitch + g];
#pragma unroll
        for (int kk = 0; kk < kGfBK / 4; ++kk) {
            const float a0 = AKM ? sA[buf][(kk * 4 + g) * kPitchK + wm * 32 + i16] : la[kk * 4];
            const float a1 = AKM ? sA[buf][(kk * 4 + g) * kPitchK + wm * 32 + 16 + i16] : la[16 * kGfPitch + kk * 4];
            const float b0 = BKM ? sB[buf][(kk * 4 + g) * kPitchK + wn * 32 + i16] : lb[kk * 4];
            const float b1 = BKM ? sB[buf][(kk * 4 + g) * kPitchK + wn * 32 + 16 + i16] : lb[16 * kGfPitch + kk * 4];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        // (the write of stage kt + 2 into this buffer is ordered behind the barrier of stage kt + 1)
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 32 + b * 16 + i16;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 32 + a * 16 + 4 * g + i;
                if (m < M) Cz[(int64_t)m * ldc + n] = gf_act(alpha * acc[a][b][i] + bv, act);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second generation for the large products (1x1 convolutions of config #2 and their weight gradients: M, N >= 128).
// The 64 x 64 kernel above tops out at ~80 TFLOP/s (half of the 157 TFLOP/s f32 matrix rate): 16 MFMAs of 32 cycles per wave
// between two barriers do not cover the stage's loads, LDS writes and barrier.  Here a workgroup owns a 128 x 128 tile, a wave a
// 64 x 64 quadrant as 2 x 2 v_mfma_f32_32x32x2_f32 tiles (64 accumulator registers): 32 MFMAs of 64 cycles per wave and stage,
// four times the matrix work per barrier, and an operand register feeds two MFMAs.
//   v_mfma_f32_32x32x2_f32: A lane (row l % 32, k l / 32), B lane (column l % 32, k l / 32), D register v of lane l = row
//   8 (v / 4) + 4 (l / 32) + v % 4 of column l % 32.
//   The 16 k of a stage are dealt to the lane halves in runs of four (half h of group q takes k = 8 q + 4 h .. + 3 - a sum over
//   k does not care about the order as long as both operands agree), so a row-major operand is read from LDS with ONE
//   ds_read_b128 per four MFMAs (rows padded to 20 floats: the 16 rows of a quarter wave fall on the 16 distinct 16-byte bank
//   slots); a K-major operand is read element-wise (consecutive lanes = consecutive columns).
typedef __attribute__((ext_vector_type(16))) float gf_f32x16;
constexpr int kGbBM = 128, kGbBN = 128, kGbPitch = 20, kGbPitchK = 132, kGbOp = 128 * kGbPitch;

template <bool AKM, bool BKM>
__global__ __launch_bounds__(kGfThreads, 2) void gemm_f32_big_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                    const float *__restrict__ bias, float *__restrict__ C, int M,
                                                                    int N, int K, int lda, int ldb, int ldc, int64_t sa, int64_t sb,
                                                                    int64_t sc, int splits, int chunk, float alpha, int act, int nt_n,
                                                                    int ntiles, int tiles_per_xcd) {
    __shared__ __attribute__((aligned(16))) float sA[2][kGbOp];
    __shared__ __attribute__((aligned(16))) float sB[2][kGbOp];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (workgroup L runs on XCD L % 8): every XCD owns a contiguous run of tiles, the tiles of one
    // column block (the same columns of B with every row block of A) next to each other in it
    const int tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int nt_m = ntiles / nt_n;
    const int tn = tile / nt_m, tm = tile - tn * nt_m;
    const int z = blockIdx.y, zb = z / splits, zs = z - zb * splits;
    const int m0 = tm * kGbBM, n0 = tn * kGbBN;
    const int kbeg = zs * chunk, kend = min(K, kbeg + chunk);
    const float *Az = A + zb * sa, *Bz = B + zb * sb;
    float *Cz = C + (int64_t)z * sc;
    const bool vec_a = (lda & 3) == 0 && ((size_t)Az & 15) == 0 && (AKM || (kbeg & 3) == 0);
    const bool vec_b = (ldb & 3) == 0 && ((size_t)Bz & 15) == 0 && (BKM || (kbeg & 3) == 0);
    // staging, row-major operand: rows tid / 4 and + 64, floats 4 (tid % 4) .. + 3 of the stage;
    // K-major operand: k rows tid / 32 and + 8 of the stage, columns 4 (tid % 32) .. + 3 of the tile
    const int lr = tid >> 2, lq = (tid & 3) * 4;
    const int kr = tid >> 5, nq = (tid & 31) * 4;
    auto load_rm = [&](const float *base, int ld, int row, int rmax, int k, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rmax && k < kend) {
            const float *p = base + (int64_t)row * ld + k;
            if (vec && k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);
            else {
                v.x = p[0];
                if (k + 1 < kend) v.y = p[1];
                if (k + 2 < kend) v.z = p[2];
                if (k + 3 < kend) v.w = p[3];
            }
        }
        return v;
    };
    auto load_km = [&](const float *base, int ld, int c0, int cmax, int k, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n = c0 + nq;
        if (k < kend && n < cmax) {
            const float *p = base + (int64_t)k * ld + n;
            if (vec && n + 3 < cmax) v = *reinterpret_cast<const float4 *>(p);
            else {
                v.x = p[0];
                if (n + 1 < cmax) v.y = p[1];
                if (n + 2 < cmax) v.z = p[2];
                if (n + 3 < cmax) v.w = p[3];
            }
        }
        return v;
    };
    float4 pa[2], pb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pa[j] = AKM ? load_km(Az, lda, m0, M, k0 + kr + 8 * j, vec_a) : load_rm(Az, lda, m0 + lr + 64 * j, M, k0 + lq, vec_a);
            pb[j] = BKM ? load_km(Bz, ldb, n0, N, k0 + kr + 8 * j, vec_b) : load_rm(Bz, ldb, n0 + lr + 64 * j, N, k0 + lq, vec_b);
        }
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    gf_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    const int nk = (kend - kbeg + 15) / 16;
    fetch(kbeg);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (AKM) *reinterpret_cast<float4 *>(&sA[buf][(kr + 8 * j) * kGbPitchK + nq]) = pa[j];
            else *reinterpret_cast<float4 *>(&sA[buf][(lr + 64 * j) * kGbPitch + lq]) = pa[j];
            if (BKM) *reinterpret_cast<float4 *>(&sB[buf][(kr + 8 * j) * kGbPitchK + nq]) = pb[j];
            else *reinterpret_cast<float4 *>(&sB[buf][(lr + 64 * j) * kGbPitch + lq]) = pb[j];
        }
        __syncthreads();
        if (kt + 1 < nk) fetch(kbeg + (kt + 1) * 16);                // in flight during the MFMAs below
        const float *la = AKM ? &sA[buf][(4 * h) * kGbPitchK + wm * 64 + r] : &sA[buf][(wm * 64 + r) * kGbPitch + 4 * h];
        const float *lb = BKM ? &sB[buf][(4 * h) * kGbPitchK + wn * 64 + r] : &sB[buf][(wn * 64 + r) * kGbPitch + 4 * h];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float av[2][4], bv[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (AKM) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[t][i] = la[(8 * q + i) * kGbPitchK + 32 * t];
                } else {
                    const float4 v = *reinterpret_cast<const float4 *>(la + 32 * t * kGbPitch + 8 * q);
                    av[t][0] = v.x; av[t][1] = v.y; av[t][2] = v.z; av[t][3] = v.w;
                }
                if (BKM) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bv[t][i] = lb[(8 * q + i) * kGbPitchK + 32 * t];
                } else {
                    const float4 v = *reinterpret_cast<const float4 *>(lb + 32 * t * kGbPitch + 8 * q);
                    bv[t][0] = v.x; bv[t][1] = v.y; bv[t][2] = v.z; bv[t][3] = v.w;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][i], bv[0][i], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][i], bv[1][i], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][i], bv[0][i], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][i], bv[1][i], acc[1][1], 0, 0, 0);
            }
        }
        // (the write of stage kt + 2 into this buffer is ordered behind the barrier of stage kt + 1)
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + r;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + a * 32 + 8 * (v >> 2) + 4 * h + (v & 3);
                if (m < M) Cz[(int64_t)m * ldc + n] = gf_act(alpha * acc[a][b][v] + bv, act);
            }
        }
    }
}


// Column sums of a token-stream gradient dY [M, N] (the bias gradient of an fp32 linear), optionally behind the ReLU mask of the
// layer's output (dYm = dY * (y > 0), written for the two gradient GEMMs): per-split partial rows part[split][N] for the step's
// deferred reduction.  ATen's sum over dim 0 takes ~20 us for 8 MB (256 output columns = little parallelism) and the mask is a
// compare + multiply pair of launches; this is one pass at copy speed.  Block = 256 threads = (256 / (N / 4)) row lanes x N / 4
// four-column groups over `rows` consecutive rows.
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float *__restrict__ d, const float *__restrict__ y, float *__restrict__ dm,
                                                         float *__restrict__ part, int M, int N, int rows) {
    __shared__ float4 red[256];
    const int ng = N >> 2, RL = 256 / ng;
    const int rl = threadIdx.x / ng, cg = threadIdx.x - rl * ng;
    const int m0 = blockIdx.x * rows, m1 = min(M, m0 + rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < RL) {
        for (int m = m0 + rl; m < m1; m += RL) {
            const int64_t o = (int64_t)m * N + 4 * cg;
            float4 v = *reinterpret_cast<const float4 *>(d + o);
            if (y) {
                const float4 t = *reinterpret_cast<const float4 *>(y + o);
                v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4 *>(dm + o) = v;
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int r = 1; r < RL; ++r) {
            const float4 t = red[r * ng + cg];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        *reinterpret_cast<float4 *>(part + (int64_t)blockIdx.x * N + 4 * cg) = acc;
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

// C[z] (z = batch * splits + split, stride sc) = act(alpha * A[batch](:, k range of the split) . B[batch](:, same range)^T + bias).
// splits > 1: the K range is cut into chunks of `chunk` (a multiple of 4) and every z writes its own partial product (bias /
// act are then the caller's business: pass NULL / 0).  Strides in elements; sa / sb = 0 shares an operand between the batches.
static int gemm_f32_launch(int akm, int bkm, const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K,
                           int lda, int ldb, int ldc, int64_t sa, int64_t sb, int64_t sc, int splits, int chunk, float alpha, int act,
                           void *stream) {
    if (batch == 0 || M == 0 || N == 0) return DFINE_OK;
    if (!A || !B || !C || batch < 0 || M < 0 || N < 0 || K < 1 || lda < 1 || ldb < 1 || ldc < N || splits < 1 || act < 0 || act > 3 ||
        (splits > 1 && (chunk < 4 || (chunk & 3) || (int64_t)chunk * (splits - 1) >= K)) || (int64_t)batch * splits > 65535)
        return DFINE_E_BADARG;
    const int ch = splits > 1 ? chunk : K;
    hipStream_t st = (hipStream_t)stream;
    // large products: 128 x 128 tiles when they still make at least ~one workgroup per CU
    const int bt_n = (N + kGbBN - 1) / kGbBN, bt_m = (M + kGbBM - 1) / kGbBM;
    if (M >= 96 && N >= 96 && (int64_t)bt_n * bt_m * batch * splits >= 224) {
        const int ntiles = bt_n * bt_m, per = (ntiles + 7) / 8;
        const dim3 gridb(8 * per, batch * splits);
#define DFINE_GB(AK, BK) hipLaunchKernelGGL((gemm_f32_big_kernel<AK, BK>), gridb, dim3(kGfThreads), 0, st, A, B, bias, C, M, N, K, lda, ldb, \
                                            ldc, sa, sb, sc, splits, ch, alpha, act, bt_n, ntiles, per)
        if (akm && bkm) DFINE_GB(true, true);
        else if (akm) DFINE_GB(true, false);
        else if (bkm) DFINE_GB(false, true);
        else DFINE_GB(false, false);
#undef DFINE_GB
        return check_launch();
    }
    const int nt_n = (N + kGfBN - 1) / kGfBN, nt_m = (M + kGfBM - 1) / kGfBM;
    const dim3 grid(nt_n * nt_m, batch * splits);
#define DFINE_GF(AK, BK) hipLaunchKernelGGL((gemm_f32_nt_kernel<AK, BK>), grid, dim3(kGfThreads), 0, st, A, B, bias, C, M, N, K, lda, ldb, ldc, \
                                            sa, sb, sc, splits, ch, alpha, act, nt_n)
    if (akm && bkm) DFINE_GF(true, true);
    else if (akm) DFINE_GF(true, false);
    else if (bkm) DFINE_GF(false, true);
    else DFINE_GF(false, false);
#undef DFINE_GF
    return check_launch();
}

int dfine_gemm_f32_nt(const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K, int lda, int ldb,
                      int ldc, int64_t sa, int64_t sb, int64_t sc, int splits, int chunk, float alpha, int act, void *stream) {
    return gemm_f32_launch(0, 0, A, B, bias, C, batch, M, N, K, lda, ldb, ldc, sa, sb, sc, splits, chunk, alpha, act, stream);
}

// The same with B given K-major: C[z][M, N] = act(alpha * A[b][M, K] . B[b][K, N] + bias[N]); B rows (ldb) are N-contiguous.
// (fp32 1x1 convolution on NCHW maps: y[b] = W x[b] and dx[b] = W^T dy[b] with N = H * W.)
int dfine_gemm_f32_nn(const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K, int lda, int ldb,
                      int ldc, int64_t sa, int64_t sb, int64_t sc, float alpha, int act, void *stream) {
    return gemm_f32_launch(0, 1, A, B, bias, C, batch, M, N, K, lda, ldb, ldc, sa, sb, sc, 1, K, alpha, act, stream);
}

// General form: a_kmajor / b_kmajor say which operands are stored K-major ([K, M] / [K, N]); splits as dfine_gemm_f32_nt.
int dfine_gemm_f32(int a_kmajor, int b_kmajor, const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K,
                   int lda, int ldb, int ldc, int64_t sa, int64_t sb, int64_t sc, int splits, int chunk, float alpha, int act, void *stream) {
    return gemm_f32_launch(a_kmajor, b_kmajor, A, B, bias, C, batch, M, N, K, lda, ldb, ldc, sa, sb, sc, splits, chunk, alpha, act, stream);
}


// number of partial rows dfine_colsum_f32 writes for M rows
int dfine_colsum_f32_splits(int M) {
    int sp = (M + 31) / 32;
    return sp > 256 ? 256 : (sp < 1 ? 1 : sp);
}

// part [splits][N] = per-split column sums of d [M, N] (N % 4 == 0, 4 <= N <= 1024), or - with relu_y [M, N] given - of
// dm = d * (relu_y > 0), which is written as well.  The bias gradient of nn.Linear (+ the ReLU backward of the MLP layers,
// ref dfine_decoder.py:33-46) in one pass.
int dfine_colsum_f32(const float *d, const float *relu_y, float *dm, float *part, int M, int N, void *stream) {
    if (M == 0) return DFINE_OK;
    if (!d || !part || M < 0 || N < 4 || N > 1024 || (N & 3) || (relu_y && !dm)) return DFINE_E_BADARG;
    const int splits = dfine_colsum_f32_splits(M);
    const int rows = (M + splits - 1) / splits;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(splits), dim3(256), 0, (hipStream_t)stream, d, relu_y, dm, part, M, N, rows);
    return check_launch();
}

}  // extern "C"
