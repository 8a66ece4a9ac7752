// Where the 128 x 128 fp32 GEMM loses time: the product kernel (PROBE 0), without the global loads of the loop (1), without
// the MFMAs (2).   hipcc --offload-arch=gfx950 -O3 -w tools/probe/gemm_f32_probe.hip -o tools/probe/gemm_f32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
namespace dfine {
constexpr int kGfThreads = 256;
__device__ __forceinline__ float gf_act(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : v; }
typedef __attribute__((ext_vector_type(16))) float gf_f32x16;
constexpr int kGbBM = 128, kGbBN = 128, kGbPitch = 20, kGbPitchK = 132, kGbOp = 128 * kGbPitch;

template <bool AKM, bool BKM, int PROBE>
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
        if (PROBE != 3) __syncthreads();
        if (PROBE != 1 && kt + 1 < nk) fetch(kbeg + (kt + 1) * 16);                // in flight during the MFMAs below
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
            for (int i = 0; i < (PROBE == 2 ? 0 : 4); ++i) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][i], bv[0][i], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][i], bv[1][i], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][i], bv[0][i], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][i], bv[1][i], acc[1][1], 0, 0, 0);
            }
        }
        if (PROBE == 2) { acc[0][0][0] += 0.f; }
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

}
using namespace dfine;
template <bool AK, bool BK, int P> void run(const char *name, int batch, int M, int N, int K, int64_t sa, int64_t sb, int lda, int ldb) {
    float *A, *B, *C;
    hipMalloc(&A, sizeof(float) * (sa ? sa * batch : (int64_t)M * K)); hipMalloc(&B, sizeof(float) * (sb ? sb * batch : (int64_t)N * K));
    hipMalloc(&C, sizeof(float) * (int64_t)batch * M * N);
    hipMemset(A, 0, sizeof(float) * (sa ? sa * batch : (int64_t)M * K)); hipMemset(B, 0, sizeof(float) * (sb ? sb * batch : (int64_t)N * K));
    const int bt_n = (N + 127) / 128, bt_m = (M + 127) / 128, ntiles = bt_n * bt_m, per = (ntiles + 7) / 8;
    dim3 grid(8 * per, batch);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r)
            hipLaunchKernelGGL((gemm_f32_big_kernel<AK, BK, P>), grid, dim3(256), 0, 0, A, B, (const float *)nullptr, C, M, N, K, lda, ldb, N, sa, sb,
                               (int64_t)M * N, 1, K, 1.f, 0, bt_n, ntiles, per);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-40s probe %d: %8.1f us  %6.1f TFLOP/s-equivalent\n", name, P, ms * 1e3, 2.0 * batch * M * N * K / ms / 1e9);
    hipFree(A); hipFree(B); hipFree(C);
}
int main() {
    // y[b] = W x[b]: A = W [512, 512] shared row-major, B = x[b] [512, 6400] K-major
    run<false, true, 0>("conv1x1 512->512 @80x80 b16 (NN)", 16, 512, 6400, 512, 0, 512 * 6400, 512, 6400);
    run<false, true, 1>("conv1x1 512->512 @80x80 b16 (NN)", 16, 512, 6400, 512, 0, 512 * 6400, 512, 6400);
    run<false, true, 2>("conv1x1 512->512 @80x80 b16 (NN)", 16, 512, 6400, 512, 0, 512 * 6400, 512, 6400);
    run<false, true, 3>("conv1x1 512->512 @80x80 b16 (NN)", 16, 512, 6400, 512, 0, 512 * 6400, 512, 6400);
    // token stream: [134400, 256] x [256, 256]^T
    run<false, false, 0>("linear 134400 x 256 -> 256 (NT)", 1, 134400, 256, 256, 0, 0, 256, 256);
    run<false, false, 1>("linear 134400 x 256 -> 256 (NT)", 1, 134400, 256, 256, 0, 0, 256, 256);
    run<false, false, 2>("linear 134400 x 256 -> 256 (NT)", 1, 134400, 256, 256, 0, 0, 256, 256);
    run<false, false, 3>("linear 134400 x 256 -> 256 (NT)", 1, 134400, 256, 256, 0, 0, 256, 256);
    return 0;
}
