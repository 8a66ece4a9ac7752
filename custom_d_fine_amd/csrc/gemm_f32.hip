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
    const int nt_n = (N + kGfBN - 1) / kGfBN, nt_m = (M + kGfBM - 1) / kGfBM;
    const dim3 grid(nt_n * nt_m, batch * splits);
    const int ch = splits > 1 ? chunk : K;
    hipStream_t st = (hipStream_t)stream;
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

}  // extern "C"
