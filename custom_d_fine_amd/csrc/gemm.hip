// A2 (AIFI) / A3 / A5 / A6 - token-stream linear layers on the MFMA units:
//     Y[M, N] = act(X[M, K] . W[N, K]^T + bias[N])            bf16 operands, fp32 accumulate, bf16 (or fp32) output
// Reference call sites: nn.Linear inside MLP / FFN / Gate / attention projections / enc_output / score heads
// (src/d_fine/arch/dfine_decoder.py:33-46,119-178,214-271,828-873, src/d_fine/arch/hybrid_encoder.py:243-290) - each an
// ATen addmm (hipBLASLt) plus separate activation kernels.  M = B * Lq = 15 744 token rows (12 800 for AIFI, 268 800 for
// the encoder-output heads), N, K <= 1024: small GEMMs whose cost is launch + one pass over X and Y, so the kernel is one
// launch with the bias / ReLU / GELU / SiLU epilogue fused and nothing else around it.
// The data gradient dX[M, K] = dY[M, N] . W[N, K] is the same kernel on a bf16 TRANSPOSED copy of the weight
// (dfine_multi_cast_bf16_t refreshes all transposed shadows with one launch per optimizer step).
//
// Both operands are K-contiguous, so MFMA fragments are plain 16-byte LDS reads:
//   block = 256 threads = 4 waves (2 x 2), tile 128 (n) x 128 (m) x 64 (k); wave = 64 x 64 = 4 x 4 MFMA 16x16x32 tiles.
//   A operand = W rows (n), B operand = X rows (m)  ->  D lane l = Y[m = l & 15][n = 4 (l >> 4) .. + 3]: four consecutive
//   output features per lane, one 8-byte store (the 4 lanes sharing an m write 32 contiguous bytes).
//   LDS rows are 72 elements (144 B): the 16 rows of a fragment read fall into 16 different 16-byte bank slots.
//   Register-staged double buffering, one barrier per 64-deep k step (the write of stage t + 2 is ordered behind the
//   barrier of stage t + 1, which every wave reaches only after its MFMAs of stage t).
//   XCD-aware block order: the n tiles of one m tile are consecutive workgroups of ONE XCD (X tile read from HBM once).
#include "common.h"

namespace dfine {

typedef __attribute__((ext_vector_type(8))) __bf16 g_bf16x8;
typedef __attribute__((ext_vector_type(4))) float g_f32x4;

constexpr int kGemmThreads = 256;
constexpr int kGBN = 128, kGBK = 64, kGPitch = 72;

constexpr int kActMask = 5;      // "activation" code of the data-gradient form: Y = (X W^T) where aux[m][n] > 0, else 0 (aux rides in `bias`)

// bf16 value > 0 (what ATen's threshold_backward tests on the saved ReLU output): not NaN-aware beyond "sign clear, non-zero"
__device__ __forceinline__ bool relu_mask_keep(uint16_t bits) { return bits != 0 && !(bits & 0x8000u) && !((bits & 0x7f80u) == 0x7f80u && (bits & 0x7fu)); }

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));      // exact GELU (nn.GELU default)
    if (act == 3) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    return v;
}

// x [M, K] (ld = ldx), w [N, K] (ld = ldw) bf16; bias fp32 [N] or null; y [M, N] (ld = ldy) bf16 (OUT32 = 0) / fp32 (1).
template <int ACT, int OUT32, int MT>                 // MT = 16-row m tiles per wave: block tile = 128 (n) x 32 MT (m)
__global__ __launch_bounds__(kGemmThreads) void linear_act_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w,
                                                                  const float *__restrict__ bias, void *__restrict__ yv,
                                                                  int M, int N, int K, int ldx, int ldw, int ldy, int nt_n,
                                                                  int nt_m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint16_t *sA = reinterpret_cast<uint16_t *>(lds);                       // [2][kGBN][kGPitch]  W rows
    constexpr int BM = 32 * MT;
    uint16_t *sB = sA + 2 * kGBN * kGPitch;                                 // [2][BM][kGPitch]  X rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tn = slot % nt_n, tm = (slot / nt_n) * 8 + xcd;
    if (tm >= nt_m) return;
    const int n0 = tn * kGBN, m0 = tm * BM;
    const int wn = wave >> 1, wm = wave & 1;
    const int g = lane >> 4, i16 = lane & 15;

    g_f32x4 acc[4][MT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = g_f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: 128 rows x 8 sixteen-byte chunks per operand = 1024 chunks / 256 threads = 4 per thread and operand
    const int st_chunk = tid & 7, st_row = tid >> 3;                        // rows st_row + 32 j
    const bool vec_x = (ldx & 7) == 0 && (K & 7) == 0, vec_w = (ldw & 7) == 0 && (K & 7) == 0;
    uint4 pa[4], pb[MT];
    auto load_rows = [&](const uint16_t *base, int ld, int r0, int rmax, bool vec, int k0, uint4 *dst, int nj) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= nj) break;
            const int r = r0 + st_row + 32 * j, k = k0 + st_chunk * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < rmax && k < K) {
                const uint16_t *p = base + (int64_t)r * ld + k;
                if (vec) v = *reinterpret_cast<const uint4 *>(p);          // K % 8 == 0: the chunk is entirely inside the row
                else {                                                     // odd K (4, 20, 33 ...): element loads
                    uint32_t q0 = 0u, q1 = 0u, q2 = 0u, q3 = 0u;
                    if (k + 0 < K) q0 |= (uint32_t)p[0];
                    if (k + 1 < K) q0 |= (uint32_t)p[1] << 16;
                    if (k + 2 < K) q1 |= (uint32_t)p[2];
                    if (k + 3 < K) q1 |= (uint32_t)p[3] << 16;
                    if (k + 4 < K) q2 |= (uint32_t)p[4];
                    if (k + 5 < K) q2 |= (uint32_t)p[5] << 16;
                    if (k + 6 < K) q3 |= (uint32_t)p[6];
                    if (k + 7 < K) q3 |= (uint32_t)p[7] << 16;
                    v = make_uint4(q0, q1, q2, q3);
                }
            }
            dst[j] = v;
        }
    };
    auto fetch = [&](int k0) {
        load_rows(w, ldw, n0, N, vec_w, k0, pa, 4);
        load_rows(x, ldx, m0, M, vec_x, k0, pb, MT);
    };
    auto stash = [&](int buf) {
        uint16_t *da = sA + buf * kGBN * kGPitch, *db = sB + buf * BM * kGPitch;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4 *>(da + (st_row + 32 * j) * kGPitch + st_chunk * 8) = pa[j];
#pragma unroll
        for (int j = 0; j < MT; ++j) *reinterpret_cast<uint4 *>(db + (st_row + 32 * j) * kGPitch + st_chunk * 8) = pb[j];
    };

    const int nk = (K + kGBK - 1) / kGBK;
    fetch(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        stash(buf);
        __syncthreads();
        if (kt + 1 < nk) fetch((kt + 1) * kGBK);                            // in flight during the MFMAs below
        const uint16_t *la = sA + buf * kGBN * kGPitch + (wn * 64 + i16) * kGPitch + 8 * g;
        const uint16_t *lb = sB + buf * BM * kGPitch + (wm * 16 * MT + i16) * kGPitch + 8 * g;
#pragma unroll
        for (int ks = 0; ks < kGBK / 32; ++ks) {
            g_bf16x8 af[4], bf[MT];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[t] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4 *>(la + t * 16 * kGPitch + ks * 32));
#pragma unroll
            for (int t = 0; t < MT; ++t)
                bf[t] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4 *>(lb + t * 16 * kGPitch + ks * 32));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }

    // ---- epilogue: bias + activation, lane -> Y[m][n .. n + 3] ------------------------------------------------------
    const bool vec_y = OUT32 ? (ldy & 3) == 0 : (ldy & 3) == 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = n0 + wn * 64 + a * 16 + 4 * g;
        if (n >= N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (ACT != kActMask && bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < N) bv[r] = bias[n + r];
        }
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = m0 + wm * 16 * MT + b * 16 + i16;
            if (m >= M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[a][b][r] + bv[r], ACT);
            if (ACT == kActMask) {                       // ReLU backward by the saved output: keep where aux[m][n] > 0
                const uint16_t *ap = reinterpret_cast<const uint16_t *>(bias) + (int64_t)m * ldy + n;
                uint16_t av[4] = {0, 0, 0, 0};
                if (vec_y && n + 3 < N) {
                    const uint2 q = *reinterpret_cast<const uint2 *>(ap);
                    av[0] = q.x & 0xffffu; av[1] = q.x >> 16; av[2] = q.y & 0xffffu; av[3] = q.y >> 16;
                } else {
                    for (int r = 0; r < 4 && n + r < N; ++r) av[r] = ap[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = relu_mask_keep(av[r]) ? v[r] : 0.f;
            }
            if (OUT32) {
                float *yp = reinterpret_cast<float *>(yv) + (int64_t)m * ldy + n;
                if (vec_y && n + 3 < N) *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                else for (int r = 0; r < 4 && n + r < N; ++r) yp[r] = v[r];
            } else {
                uint16_t *yp = reinterpret_cast<uint16_t *>(yv) + (int64_t)m * ldy + n;
                if (vec_y && n + 3 < N) {
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(yp) = o;
                } else {
                    for (int r = 0; r < 4 && n + r < N; ++r) yp[r] = f32_to_bf16(v[r]);
                }
            }
        }
    }
}

// Second generation for K % 64 == 0 and 16-byte aligned rows (every projection / FFN / head-trunk layer of the token
// streams): the same tile product fed by asynchronous global -> LDS copies into a ring, like conv1x1_glds_kernel.  The
// register-staged kernel above has ONE k step of loads in flight per workgroup and exposes a full L2 / HBM round trip per
// 64-deep step (16 - 32 MFMAs of work): 0.2 PFLOP/s on 15 744 x 1024 x 256, 168 us on 268 800 x 256 x 256.
//   stage  = 64 k: W rows [128][64] + X rows [64 MT][64], 8-row x 128-byte pieces copied by LDS-DMA; row r keeps its
//            16-byte k chunk c at chunk c ^ (r & 7) (the copy writes LDS lane-linearly, so the swizzle is applied on the
//            SOURCE side; the 16 lanes of one ds_read_b128 group then hit 16 different bank slots).
//   ring   = R stages (4 for MT 2: a whole K = 256 row block is in flight at once; 3 for MT 1 / 4), counted vmcnt + ONE
//            barrier per stage; the copies of stage s + R - 1 are issued behind the fragment reads of stage s.
//   block  = 512 threads = 8 waves (2 over n x 4 over m), tile 128 (n) x 64 MT (m), wave = 64 (n) x 16 MT (m).
constexpr int kLRThreads = 512;

__device__ __forceinline__ void lr_glds16(const uint16_t *gsrc, unsigned lds_addr) {      // see glds16 in conv.hip
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

template <int ACT, int OUT32, int MT, int R>
__global__ __launch_bounds__(kLRThreads) void linear_ring_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w,
                                                                 const float *__restrict__ bias, void *__restrict__ yv,
                                                                 int M, int N, int K, int ldx, int ldw, int ldy, int nt_n,
                                                                 int nt_m, int total_v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int BM = 64 * MT, ROWS = kGBN + BM, SB = ROWS * 128, PPW = ROWS / 64;   // 8-row pieces per wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    const int g = lane >> 4, i16 = lane & 15;
    const int G = gridDim.x;                                               // a multiple of 8: a workgroup stays on its XCD
    const int nk = K / kGBK;

    // virtual tile index v -> (tn, tm): the n tiles of one m tile are consecutive slots of ONE XCD (v & 7)
    auto decode = [&](int v, int &n0, int &m0) -> bool {
        const int xcd = v & 7, slot = v >> 3;
        const int tn = slot % nt_n, tm = (slot / nt_n) * 8 + xcd;
        n0 = tn * kGBN; m0 = tm * BM;
        return tm < nt_m;
    };
    auto next_valid = [&](int v) {
        int n0, m0;
        while (v < total_v && !decode(v, n0, m0)) v += G;
        return v;
    };

    // ---- copy side: a flat stream of stages over this workgroup's tiles; the ring keeps filling across tile boundaries, so
    // the loads of the next tile are in flight during the last MFMAs and the stores of the current one ----
    // LDS image of a stage: row r keeps its 16-byte k chunk c at chunk c ^ f(r).  X rows: f = r & 7.  W rows: the MFMA "row" i
    // of A tile a is W row 16 (i >> 2) + 4 a + (i & 3) of the wave's 64 (below), f = 2 ((r >> 4) & 3) + ((r >> 1) & 1): the 16
    // lanes of one ds_read_b128 group then hit 16 different bank slots in both regions.
    const int prow = lane >> 3, pc = lane & 7;
    int iv = next_valid(blockIdx.x), is_ = 0, qi = 0;
    const uint16_t *src[PPW];
    auto set_src = [&]() {
        int n0, m0;
        decode(iv, n0, m0);
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = (wave * PPW + j) * 8 + prow;
            if (row < kGBN) src[j] = w + (int64_t)min(n0 + row, N - 1) * ldw + ((pc ^ (2 * ((row >> 4) & 3) + ((row >> 1) & 1))) << 3);
            else src[j] = x + (int64_t)min(m0 + row - kGBN, M - 1) * ldx + ((pc ^ (row & 7)) << 3);   // rows past the matrix: any
        }                                                                                            // valid row, never stored
    };
    if (iv < total_v) set_src();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + wave * PPW * 1024;
    auto issue_next = [&]() {
        if (iv >= total_v) return;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (qi % R) * SB);
#pragma unroll
        for (int j = 0; j < PPW; ++j) lr_glds16(src[j] + is_ * kGBK, __builtin_amdgcn_readfirstlane(base + j * 1024));
        ++qi;
        if (++is_ == nk) {
            is_ = 0;
            iv = next_valid(iv + G);
            if (iv < total_v) set_src();
        }
    };
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue_next();

    // A tile a, MFMA row i16 -> W row 16 (i16 >> 2) + 4 a + (i16 & 3): lane (i16, g) then ends up with the 16 CONSECUTIVE
    // output features 16 g .. 16 g + 15 of token row i16 (a = 0 .. 3, r = 0 .. 3) - 32-byte pieces of whole 128-byte rows
    const int ar = wn * 64 + 16 * (i16 >> 2) + (i16 & 3), fa = 2 * (i16 >> 2) + ((i16 >> 1) & 1);
    const int arow = ar * 128, brow = (kGBN + wm * 16 * MT + i16) * 128;
    const int sa0 = (g ^ fa) << 4, sa1 = ((4 + g) ^ fa) << 4;
    const int sb0 = (g ^ (i16 & 7)) << 4, sb1 = ((4 + g) ^ (i16 & 7)) << 4;

    int q = 0;
    for (int cv = next_valid(blockIdx.x); cv < total_v; cv = next_valid(cv + G)) {
        g_f32x4 acc[4][MT];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = g_f32x4{0.f, 0.f, 0.f, 0.f};
        int n0, m0;
        decode(cv, n0, m0);
        const int n = n0 + wn * 64 + 16 * g;
        // the bias is fetched HERE: a load in the epilogue would be waited for with the copies of the next tile in front of it
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = (ACT != kActMask && bias && n + e < N) ? bias[n + e] : 0.f;
        for (int s = 0; s < nk; ++s, ++q) {
            // loads return in order: "at most (younger copies) outstanding" means this stage has landed (stores of the
            // previous tile may still be counted - the wait is then longer than needed, never shorter)
            const int ahead = qi - q - 1;
            if (R >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char *st = lds + (q % R) * SB;
            g_bf16x8 af[2][4], bf[2][MT];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[0][t] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4 *>(st + arow + t * 512 + sa0));
                af[1][t] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4 *>(st + arow + t * 512 + sa1));
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                bf[0][t] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4 *>(st + brow + t * 2048 + sb0));
                bf[1][t] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4 *>(st + brow + t * 2048 + sb1));
            }
            issue_next();                                // into the slot of stage q - 1: every wave is past it (the barrier above)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < MT; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[h][a], bf[h][b], acc[a][b], 0, 0, 0);
        }

        // ---- epilogue: bias + activation; lane -> Y[m][n .. n + 15] ----
        if (n >= N) continue;
        const bool full = n + 15 < N;
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = m0 + wm * 16 * MT + b * 16 + i16;
            if (m >= M) continue;
            float v[16];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * a + r] = act_apply(acc[a][b][r] + bv[4 * a + r], ACT);
            if (ACT == kActMask) {                       // ReLU backward by the saved output: keep where aux[m][n] > 0
                const uint16_t *ap = reinterpret_cast<const uint16_t *>(bias) + (int64_t)m * ldy + n;
                if (full && (ldy & 7) == 0) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint4 q = reinterpret_cast<const uint4 *>(ap)[e];
                        const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (!relu_mask_keep((uint16_t)(qw[j] & 0xffffu))) v[8 * e + 2 * j] = 0.f;
                            if (!relu_mask_keep((uint16_t)(qw[j] >> 16))) v[8 * e + 2 * j + 1] = 0.f;
                        }
                    }
                } else {
                    for (int e = 0; e < 16 && n + e < N; ++e) if (!relu_mask_keep(ap[e])) v[e] = 0.f;
                }
            }
            if (OUT32) {
                float *yp = reinterpret_cast<float *>(yv) + (int64_t)m * ldy + n;
                if (full && (ldy & 3) == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reinterpret_cast<float4 *>(yp)[e] = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
                } else {
                    for (int e = 0; e < 16 && n + e < N; ++e) yp[e] = v[e];
                }
            } else {
                uint16_t *yp = reinterpret_cast<uint16_t *>(yv) + (int64_t)m * ldy + n;
                if (full && (ldy & 7) == 0) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        reinterpret_cast<uint4 *>(yp)[e] = make_uint4(pack_bf16x2(v[8 * e], v[8 * e + 1]), pack_bf16x2(v[8 * e + 2], v[8 * e + 3]),
                                                                      pack_bf16x2(v[8 * e + 4], v[8 * e + 5]), pack_bf16x2(v[8 * e + 6], v[8 * e + 7]));
                } else {
                    for (int e = 0; e < 16 && n + e < N; ++e) yp[e] = f32_to_bf16(v[e]);
                }
            }
        }
    }
}

// fp32 [rows, cols] -> bf16 [cols, rows] (transposed shadow of a Linear weight), many tensors per launch.
// table rows of 4 x int64 = {src ptr, dst ptr, rows, cols}; blockIdx.y = entry, 32 x 32 LDS tiles.
__global__ __launch_bounds__(256) void multi_cast_bf16_t_kernel(const int64_t *__restrict__ table) {
    __shared__ float tile[32][33];
    const int64_t *e = table + (int64_t)blockIdx.y * 4;
    const float *src = reinterpret_cast<const float *>(e[0]);
    uint16_t *dst = reinterpret_cast<uint16_t *>(e[1]);
    const int rows = (int)e[2], cols = (int)e[3];
    const int tr = (rows + 31) / 32, tc = (cols + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, c = c0 + tx;
            tile[ty + 8 * j][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < rows && c < cols) dst[(int64_t)c * rows + r] = f32_to_bf16(tile[tx][ty + 8 * j]);
        }
        __syncthreads();
    }
}

constexpr float kPosClamp = 10.f;       // act 4: the decoder's clamp of the query position embedding (ref dfine_decoder.py:466)

// dpre = dy * act'(.) for the fused-epilogue activations, from the saved OUTPUT y for ReLU (y > 0) and from the saved
// pre-activation z for GELU / SiLU.  bf16 in / out, 8 elements per thread.
template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ ref,
                                                      uint16_t *__restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 d = reinterpret_cast<const uint4 *>(dy)[i], r = reinterpret_cast<const uint4 *>(ref)[i];
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w}, rw[4] = {r.x, r.y, r.z, r.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g0 = __uint_as_float(dw[j] << 16), g1 = __uint_as_float(dw[j] & 0xffff0000u);
            const float z0 = __uint_as_float(rw[j] << 16), z1 = __uint_as_float(rw[j] & 0xffff0000u);
            if (ACT == 1) { g0 = z0 > 0.f ? g0 : 0.f; g1 = z1 > 0.f ? g1 : 0.f; }
            else if (ACT == 4) {                        // clamp(z, -10, 10): the gradient passes where -10 <= z <= 10 (ATen's rule)
                g0 = (z0 >= -kPosClamp && z0 <= kPosClamp) ? g0 : 0.f; g1 = (z1 >= -kPosClamp && z1 <= kPosClamp) ? g1 : 0.f;
            } else if (ACT == 2) {
                const float c = 0.70710678118654752440f, k = 0.39894228040143267794f;       // 1/sqrt(2), 1/sqrt(2 pi)
                g0 *= 0.5f * (1.f + erff(z0 * c)) + z0 * k * __expf(-0.5f * z0 * z0);
                g1 *= 0.5f * (1.f + erff(z1 * c)) + z1 * k * __expf(-0.5f * z1 * z1);
            } else if (ACT == 3) {
                const float s0 = __builtin_amdgcn_rcpf(1.f + __expf(-z0)), s1 = __builtin_amdgcn_rcpf(1.f + __expf(-z1));
                g0 *= s0 * (1.f + z0 * (1.f - s0)); g1 *= s1 * (1.f + z1 * (1.f - s1));
            }
            o[j] = pack_bf16x2(g0, g1);
        }
        reinterpret_cast<uint4 *>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// y = act(z), bf16, 8 elements per thread (training-time GELU / SiLU: the pre-activation z is what backward needs)
template <int ACT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const uint16_t *__restrict__ z, uint16_t *__restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 d = reinterpret_cast<const uint4 *>(z)[i];
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z0 = __uint_as_float(dw[j] << 16), z1 = __uint_as_float(dw[j] & 0xffff0000u);
            if (ACT == 4) o[j] = pack_bf16x2(fminf(fmaxf(z0, -kPosClamp), kPosClamp), fminf(fmaxf(z1, -kPosClamp), kPosClamp));
            else o[j] = pack_bf16x2(act_apply(z0, ACT), act_apply(z1, ACT));
        }
        reinterpret_cast<uint4 *>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace dfine

using namespace dfine;

extern "C" {

static int linear_launch(const void *x, const void *w, const float *bias, void *y, int M, int N, int K, int ldx, int ldw,
                         int ldy, int act, int out_f32, void *stream) {
    if (M == 0 || N == 0) return DFINE_OK;
    if (!x || !w || !y || M < 0 || N < 0 || K < 1 || ldx < K || ldw < K || ldy < N || act < 0 || act > kActMask || act == 4 ||
        (act == kActMask && out_f32))
        return DFINE_E_BADARG;
    const int nt_n = (N + kGBN - 1) / kGBN;
    hipStream_t st = (hipStream_t)stream;
    if ((K % kGBK) == 0 && (ldx & 7) == 0 && (ldw & 7) == 0 && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0) {
        // m tile: 256 rows once that still gives >= 4 rounds of workgroups (the 268 800-row encoder streams), 64 rows when
        // 128-row tiles would leave CUs idle, 128 otherwise
        const int64_t t128 = (int64_t)((M + 127) / 128) * nt_n;
        int mt = t128 >= 900 ? 4 : t128 >= 200 ? 2 : 1;
        const int bm = 64 * mt, nt_m = (M + bm - 1) / bm, ring = mt == 2 ? 4 : 3;
        const size_t ldsb = (size_t)ring * (kGBN + bm) * 128;
        const int total_v = 8 * ((nt_m + 7) / 8) * nt_n;
        static const int cus = [] { int d = 0, c = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c > 0 ? c : 256; }();
        const int resident = (cus & ~7) * (mt == 1 ? 2 : 1);                // workgroups that fit the chip at once (LDS bound)
        dim3 grid(total_v < resident ? total_v : resident);
        static bool ring_attr = false;
#define DFINE_LR_ATTR(A, O, T, RR)                                                                                        \
    { hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_ring_kernel<A, O, T, RR>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RR * (kGBN + 64 * T) * 128);        \
      if (r != hipSuccess) e = r; }
#define DFINE_LR_ATTR_A(O, T, RR) DFINE_LR_ATTR(0, O, T, RR) DFINE_LR_ATTR(1, O, T, RR) DFINE_LR_ATTR(2, O, T, RR) DFINE_LR_ATTR(3, O, T, RR)
        if (!ring_attr) {
            hipError_t e = hipSuccess;
            DFINE_LR_ATTR_A(0, 1, 3) DFINE_LR_ATTR_A(1, 1, 3) DFINE_LR_ATTR_A(0, 2, 4) DFINE_LR_ATTR_A(1, 2, 4)
            DFINE_LR_ATTR_A(0, 4, 3) DFINE_LR_ATTR_A(1, 4, 3)
            DFINE_LR_ATTR(kActMask, 0, 1, 3) DFINE_LR_ATTR(kActMask, 0, 2, 4) DFINE_LR_ATTR(kActMask, 0, 4, 3)
            if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
            ring_attr = true;
        }
#undef DFINE_LR_ATTR_A
#undef DFINE_LR_ATTR
#define DFINE_LR(A, O, T, RR)                                                                                             \
    hipLaunchKernelGGL((linear_ring_kernel<A, O, T, RR>), grid, dim3(kLRThreads), ldsb, st, (const uint16_t *)x,          \
                       (const uint16_t *)w, bias, y, M, N, K, ldx, ldw, ldy, nt_n, nt_m, total_v)
#define DFINE_LR_T(A, O) { if (mt == 4) DFINE_LR(A, O, 4, 3); else if (mt == 2) DFINE_LR(A, O, 2, 4); else DFINE_LR(A, O, 1, 3); }
#define DFINE_LR_O(A) { if (out_f32) DFINE_LR_T(A, 1) else DFINE_LR_T(A, 0) }
        switch (act) {
            case 0: DFINE_LR_O(0) break;
            case 1: DFINE_LR_O(1) break;
            case 2: DFINE_LR_O(2) break;
            case kActMask: DFINE_LR_T(kActMask, 0) break;
            default: DFINE_LR_O(3) break;
        }
#undef DFINE_LR_O
#undef DFINE_LR_T
#undef DFINE_LR
        return check_launch();
    }
    // 128-row m tiles when they still give >= 2 workgroups per CU; 64-row tiles otherwise (the token streams have
    // 12 800 - 15 744 rows: 100 - 123 tiles of 128 rows would leave half of the 256 CUs idle)
    const int mt = ((int64_t)((M + 127) / 128) * nt_n >= 512) ? 4 : 2;
    const int bm = 32 * mt;
    const int nt_m = (M + bm - 1) / bm;
    const size_t ldsb = (size_t)2 * (kGBN + bm) * kGPitch * 2;              // 73 728 B / 55 296 B
    dim3 grid(8 * ((nt_m + 7) / 8) * nt_n);
    static bool attr_set = false;       // once: not a stream operation, keep it out of graph capture
#define DFINE_LA_ATTR(A, O, T)                                                                                           \
    { hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_act_kernel<A, O, T>),                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kGBN + 32 * T) * kGPitch * 2); \
      if (r != hipSuccess) e = r; }
    if (!attr_set) {
        hipError_t e = hipSuccess;
        DFINE_LA_ATTR(0, 0, 4) DFINE_LA_ATTR(1, 0, 4) DFINE_LA_ATTR(2, 0, 4) DFINE_LA_ATTR(3, 0, 4)
        DFINE_LA_ATTR(0, 1, 4) DFINE_LA_ATTR(1, 1, 4) DFINE_LA_ATTR(2, 1, 4) DFINE_LA_ATTR(3, 1, 4)
        DFINE_LA_ATTR(0, 0, 2) DFINE_LA_ATTR(1, 0, 2) DFINE_LA_ATTR(2, 0, 2) DFINE_LA_ATTR(3, 0, 2)
        DFINE_LA_ATTR(0, 1, 2) DFINE_LA_ATTR(1, 1, 2) DFINE_LA_ATTR(2, 1, 2) DFINE_LA_ATTR(3, 1, 2)
        DFINE_LA_ATTR(kActMask, 0, 4) DFINE_LA_ATTR(kActMask, 0, 2)
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
#undef DFINE_LA_ATTR
#define DFINE_LA(A, O, T)                                                                                                \
    hipLaunchKernelGGL((linear_act_kernel<A, O, T>), grid, dim3(kGemmThreads), ldsb, st, (const uint16_t *)x, (const uint16_t *)w, \
                       bias, y, M, N, K, ldx, ldw, ldy, nt_n, nt_m)
#define DFINE_LA_T(A, O) { if (mt == 4) DFINE_LA(A, O, 4); else DFINE_LA(A, O, 2); }
#define DFINE_LA_O(A) { if (out_f32) DFINE_LA_T(A, 1) else DFINE_LA_T(A, 0) }
    switch (act) {
        case 0: DFINE_LA_O(0) break;
        case 1: DFINE_LA_O(1) break;
        case 2: DFINE_LA_O(2) break;
        case kActMask: DFINE_LA_T(kActMask, 0) break;
        default: DFINE_LA_O(3) break;
    }
#undef DFINE_LA_O
#undef DFINE_LA_T
#undef DFINE_LA
    return check_launch();
}

int dfine_linear_act_fwd(const void *x, const void *w, const float *bias, void *y, int M, int N, int K, int ldx, int ldw,
                         int ldy, int act, int out_f32, void *stream) {
    if (act < 0 || act > 3) return DFINE_E_BADARG;
    return linear_launch(x, w, bias, y, M, N, K, ldx, ldw, ldy, act, out_f32, stream);
}

// dX[M, N] = (dY[M, K] . W[N, K]^T) masked by a saved ReLU output: dX[m][n] = 0 where aux[m][n] <= 0 (aux bf16 [M, N], ld = ldy).
// The data gradient of the layer BEHIND a Linear + ReLU with the ReLU's backward in its store epilogue (MLP / FFN of the decoder,
// reference dfine_decoder.py:33-46,214-231: ATen runs threshold_backward as its own pass).  bf16 output; same shape rules as
// dfine_linear_act_fwd.
int dfine_linear_dgrad_relu(const void *x, const void *w, const void *aux, void *y, int M, int N, int K, int ldx, int ldw, int ldy,
                            void *stream) {
    if (!aux) return DFINE_E_BADARG;
    return linear_launch(x, w, reinterpret_cast<const float *>(aux), y, M, N, K, ldx, ldw, ldy, kActMask, 0, stream);
}

// table: device int64 [n_entries][4] = {src fp32 [rows, cols], dst bf16 [cols, rows], rows, cols}
int dfine_multi_cast_bf16_t(const void *table, int n_entries, void *stream) {
    if (n_entries == 0) return DFINE_OK;
    if (!table || n_entries < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(multi_cast_bf16_t_kernel, dim3(64, n_entries), dim3(256), 0, (hipStream_t)stream, (const int64_t *)table);
    return check_launch();
}

int dfine_act_fwd_bf16(const void *z, void *y, int64_t n, int act, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!z || !y || n < 0 || (n & 7) || act < 1 || act > 4) return DFINE_E_BADARG;
    const int64_t n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (act == 1) hipLaunchKernelGGL(act_fwd_kernel<1>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)z, (uint16_t *)y, n8);
    else if (act == 2) hipLaunchKernelGGL(act_fwd_kernel<2>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)z, (uint16_t *)y, n8);
    else if (act == 3) hipLaunchKernelGGL(act_fwd_kernel<3>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)z, (uint16_t *)y, n8);
    else hipLaunchKernelGGL(act_fwd_kernel<4>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)z, (uint16_t *)y, n8);
    return check_launch();
}

// out = dy * act'(ref): ref = saved output (act 1, ReLU) or saved pre-activation (act 2 GELU, 3 SiLU); bf16, n % 8 == 0
int dfine_act_bwd_bf16(const void *dy, const void *ref, void *out, int64_t n, int act, void *stream) {
    if (n == 0) return DFINE_OK;
    if (!dy || !ref || !out || n < 0 || (n & 7) || act < 1 || act > 4) return DFINE_E_BADARG;
    const int64_t n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (act == 1) hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)dy, (const uint16_t *)ref, (uint16_t *)out, n8);
    else if (act == 2) hipLaunchKernelGGL(act_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)dy, (const uint16_t *)ref, (uint16_t *)out, n8);
    else if (act == 3) hipLaunchKernelGGL(act_bwd_kernel<3>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)dy, (const uint16_t *)ref, (uint16_t *)out, n8);
    else hipLaunchKernelGGL(act_bwd_kernel<4>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)dy, (const uint16_t *)ref, (uint16_t *)out, n8);
    return check_launch();
}

}  // extern "C"
