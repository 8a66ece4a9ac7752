// A1 / A2 in fp32 (BASELINE configs[1]: D-FINE-s 640x640 bs 16 fp32): dense convolutions on the f32-input matrix cores,
// v_mfma_f32_16x16x4_f32 - exact fp32 products and sums (bitwise an fmaf chain, MI355X_MICROARCH.md), 157 TFLOP/s peak
// = 1/16 of the bf16 rate.  Reference call sites: the same nn.Conv2d layers as conv.hip (hgnetv2.py:35-80,115-166,
// hybrid_encoder.py:21-156), run without autocast; ATen hands them to MIOpen.
//
// One generic kernel pair instead of the bf16 family's specialised ones - with one fp32 element per lane and MFMA operand the
// layout problems of the bf16 path disappear: the B fragment of a tap is x_lds[channel k][pixel + tap shift], a plain 4-byte
// LDS read at any shift, so x is staged in its NATURAL NCHW order (no transposition, no alignment rules) and kernel sizes
// 1 / 2 / 3, strides 1 / 2 and one-sided paddings (the stem's F.pad(x, (0, 1, 0, 1)) + 2x2) are address arithmetic.
//   conv_f32_kernel   y[b, n, oy, ox] = sum_{c, ty, tx} w[n, c, ty, tx] x[b, c, oy S + ty - pt, ox S + tx - pl]
//                     block = 256 threads, (image, R output rows x CW output columns <= 160 pixels, 64 output channels);
//                     wave -> 16 output channels x all pixel tiles; 16 input channels per stage: x tile (+ halo, zero
//                     filled) and the stage's weights [tap][64][16] through LDS.
//                     MFMA: A lane l = w[n = l & 15][k = l >> 4], B lane l = x[k = l >> 4][pixel = l & 15],
//                     D lane l reg r = y[n = 4 (l >> 4) + r][pixel = l & 15].
//                     Data gradient = the same kernel on weights packed transposed + flipped (stride 2: on the zero-upsampled
//                     gradient, host side).
//   wgrad_f32_kernel  dW[n, c, ty, tx] = sum_{b, p} dy[b, n, p] x[b, c, p S + shift]: K = pixels; block = (64 n) x (16 c) x all
//                     taps for a range of (image, strip) units, fp32 partial sums per split (reduced by conv.hip's
//                     conv_wgrad_reduce_kernel, same [split][n][c][tap] layout).
#include "common.h"
#include <type_traits>

namespace dfine {

typedef __attribute__((ext_vector_type(4))) float f4v;
constexpr int kF32Threads = 256, kF32KC = 16, kF32MaxTiles = 10, kF32WP = 20;   // kF32WP: weight row pitch in LDS (16-byte rows, the
                                                                                 // 16 x 4 (n, k) reads of an A fragment on 64 distinct banks)

// fp32 master [Cout][Cin][KS][KS] -> [KS*KS][NP][KP] fp32 (NP = n rounded up to 64, KP = k rounded up to 16, zero padded).
// dgrad = 1: n = cin, k = cout, taps flipped.
__global__ void conv_pack_f32_kernel(const float *__restrict__ w, float *__restrict__ w2, int Cout, int Cin, int KS, int NP, int KP, int dgrad) {
    const int64_t total = (int64_t)KS * KS * NP * KP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KP), n = (int)((i / KP) % NP), tap = (int)(i / ((int64_t)KP * NP));
        float v = 0.f;
        if (!dgrad) {
            if (n < Cout && k < Cin) v = w[((int64_t)n * Cin + k) * KS * KS + tap];
        } else {
            const int r = tap / KS, s = tap % KS;
            const int src_tap = (KS - 1 - r) * KS + (KS - 1 - s);
            if (n < Cin && k < Cout) v = w[((int64_t)k * Cin + n) * KS * KS + src_tap];
        }
        w2[i] = v;
    }
}

// NWN = waves along the output channels (a workgroup owns 16 NWN of them), the other 4 / NWN wave groups share the pixel tiles:
// a 16-channel layer (the stem, stage 1) keeps all four waves busy on 640 pixels instead of three idle ones on 160.
// Staging issues its global loads in batches (8 x-tile loads, all weight vectors of the stage) before the LDS stores: with one
// load -> store pair per loop trip the stage was a chain of ~50 dependent memory round trips (15 us against 5 us of MFMA work).
template <int KS, int S, int NWN>
__global__ __launch_bounds__(kF32Threads, 2) void conv_f32_kernel(const float *__restrict__ x, const float *__restrict__ w2, float *__restrict__ y,
                                                               int Cin, int Cout, int NP, int KP, int Hi, int Wi, int Ho, int Wo, int pt, int pl,
                                                               int R, int CW, int strips, int ctiles, int kc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int TAPS = KS * KS, NB = 16 * NWN, PW = 4 / NWN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nw = wave % NWN, pw = wave / NWN;
    const int g = lane >> 4, i16 = lane & 15;
    int u = blockIdx.x;
    const int ct = u % ctiles; u /= ctiles;
    const int strip = u % strips; const int b = u / strips;
    const int r0 = strip * R, x0 = ct * CW;
    const int rows = min(R, Ho - r0), cols = min(CW, Wo - x0);
    const int TP = rows * cols;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS;
    const int iy0 = r0 * S - pt, ix0 = x0 * S - pl;
    const int xelems = rows_l * WL;
    float *xs = reinterpret_cast<float *>(lds_raw);                    // [kc][rows_l][WL]
    float *ws = xs + ((kc * xelems + 3) & ~3);                         // [TAPS][NB][kF32WP]
    const int n0 = blockIdx.y * NB;
    const int ntile = (TP + 15) / 16;
    const int ntw = __builtin_amdgcn_readfirstlane((ntile - pw + PW - 1) / PW);        // pixel tiles of this wave
    // LDS offset of this lane's pixel (tap 0) per pixel tile of this wave (tiles pw, pw + PW, ...)
    int pl_off[kF32MaxTiles];
#pragma unroll
    for (int jt = 0; jt < kF32MaxTiles; ++jt) {
        int q = (pw + PW * jt) * 16 + i16;
        if (q >= TP) q = 0;
        const int oy = q / cols, ox = q - oy * cols;
        pl_off[jt] = (oy * S) * WL + ox * S;
    }
    f4v acc[kF32MaxTiles];
#pragma unroll
    for (int jt = 0; jt < kF32MaxTiles; ++jt) acc[jt] = f4v{0.f, 0.f, 0.f, 0.f};
    const float *xb = x + (int64_t)b * Cin * Hi * Wi;
    constexpr int WV = TAPS * NB * 4, WIT = (WV + kF32Threads - 1) / kF32Threads;      // float4 vectors of a stage's weights
    for (int c0 = 0; c0 < KP; c0 += kF32KC) {
        const int kkn = (min(kF32KC, Cin - c0) + 3) >> 2;                  // 4-channel groups of this stage that hold data
        float4 wreg[WIT];
#pragma unroll
        for (int j = 0; j < WIT; ++j) {
            const int idx = tid + kF32Threads * j;
            wreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (WV % kF32Threads == 0 || idx < WV) {
                const int k4 = idx & 3, n = (idx >> 2) % NB, tap = idx / (4 * NB);
                wreg[j] = *reinterpret_cast<const float4 *>(w2 + ((int64_t)tap * NP + n0 + n) * KP + c0 + 4 * k4);
            }
        }
        __syncthreads();                                                   // the previous stage's MFMAs have read xs / ws
#pragma unroll
        for (int j = 0; j < WIT; ++j) {
            const int idx = tid + kF32Threads * j;
            if (WV % kF32Threads == 0 || idx < WV) {
                const int k4 = idx & 3, n = (idx >> 2) % NB, tap = idx / (4 * NB);
                *reinterpret_cast<float4 *>(ws + (tap * NB + n) * kF32WP + 4 * k4) = wreg[j];
            }
        }
        // x tile: a wave takes (channel, tile row) rows wv, wv + 4, ...; four rows x two 64-column chunks are in flight at a time
        // (the row / channel split runs on the scalar unit: an integer division pair per ELEMENT made this the bottleneck of the
        // 3-channel stem)
        const int nrows = 4 * kkn * rows_l;
        for (int rb = wv; rb < nrows; rb += 16) {
            const float *src[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rb + 4 * i;
                const int c = row / rows_l, ly = row - c * rows_l;
                const int iy = iy0 + ly;
                ok[i] = row < nrows && c0 + c < Cin && iy >= 0 && iy < Hi;
                src[i] = xb + ((int64_t)(ok[i] ? c0 + c : 0) * Hi + (ok[i] ? iy : 0)) * Wi;
            }
            for (int lx0 = 0; lx0 < WL; lx0 += 128) {
                float v[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ix = ix0 + lx0 + 64 * h + lane;
                        v[i][h] = (ok[i] && ix >= 0 && ix < Wi) ? src[i][ix] : 0.f;
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int lx = lx0 + 64 * h + lane;
                        if (rb + 4 * i < nrows && lx < WL) xs[(rb + 4 * i) * WL + lx] = v[i][h];
                    }
            }
        }
        __syncthreads();
        // the MFMA section in three sizes (2 / 5 / 10 pixel tiles per wave, chosen per workgroup): a short strip does not pay for
        // ten tiles, and every size is straight-line code on registers
        auto stage_mfma = [&](auto nt_c) {
            constexpr int NT = decltype(nt_c)::value;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int toff = (tap / KS) * WL + (tap % KS);
#pragma unroll
                for (int kk = 0; kk < kF32KC / 4; ++kk) {
                    if (kk >= kkn) break;                                  // uniform: channel groups past Cin (the stem: 3 channels)
                    const float a = ws[(tap * NB + nw * 16 + i16) * kF32WP + kk * 4 + g];
                    const float *xr = xs + (kk * 4 + g) * xelems + toff;
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)            // tiles past the strip read pixel 0 and are never stored: no branches
                        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[pl_off[jt]], acc[jt], 0, 0, 0);
                }
            }
        };
        if (ntw <= 2) stage_mfma(std::integral_constant<int, 2>{});
        else if (ntw <= 5) stage_mfma(std::integral_constant<int, 5>{});
        else stage_mfma(std::integral_constant<int, kF32MaxTiles>{});
    }
    float *yb = y + (int64_t)b * Cout * Ho * Wo;
#pragma unroll
    for (int jt = 0; jt < kF32MaxTiles; ++jt) {
        const int t = pw + PW * jt;
        const int q = t * 16 + i16;
        if (t < ntile && q < TP) {
            const int oy = q / cols, ox = q - oy * cols;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + nw * 16 + 4 * g + r;
                if (n < Cout) yb[((int64_t)n * Ho + r0 + oy) * Wo + x0 + ox] = acc[jt][r];
            }
        }
    }
}

// part[split][NP16][CP16][TAPS]; block = (16 NWN-row n tile, 16-column c tile) x split.  NWN waves along the output channels;
// the other 4 / NWN wave groups deal the pixel steps of a unit among themselves and write their own slab each
// (slab = split * (4 / NWN) + group): a 16-channel layer keeps four waves busy instead of one.  Staging in batches of 8 loads.
template <int KS, int S, int NWN>
__global__ __launch_bounds__(kF32Threads) void wgrad_f32_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part,
                                                                int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int pt, int pl, int R, int CW,
                                                                int strips, int ctiles, int total_units, int units_per_split, int nct, int NP16,
                                                                int CP16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int TAPS = KS * KS, NB = 16 * NWN, PW = 4 / NWN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nw = wave % NWN, pw = wave / NWN;
    const int g = lane >> 4, i16 = lane & 15;
    const int nt = blockIdx.x / nct, ctile = blockIdx.x - nt * nct;
    const int n0 = nt * NB, c0 = ctile * 16;
    const int split = blockIdx.y;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS, xelems = rows_l * WL;
    const int TPmax = R * CW, DP = TPmax + 4;                              // dy tile pitch
    float *xs = reinterpret_cast<float *>(lds_raw);                       // [16 c][rows_l][WL]
    float *ds = xs + 16 * xelems;                                          // [NB n][DP]
    f4v acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
    const int u0 = split * units_per_split, u1 = min(total_units, u0 + units_per_split);
    for (int uu = u0; uu < u1; ++uu) {
        int u = uu;
        const int ct = u % ctiles; u /= ctiles;
        const int strip = u % strips; const int b = u / strips;
        const int r0 = strip * R, x0 = ct * CW;
        const int rows = min(R, Ho - r0), cols = min(CW, Wo - x0);
        const int TP = rows * cols;
        const int iy0 = r0 * S - pt, ix0 = x0 * S - pl;
        const float *xb = x + (int64_t)b * Cin * Hi * Wi;
        const float *dyb = dy + (int64_t)b * Cout * Ho * Wo;
        __syncthreads();
        {                                                                  // x tile: (channel, tile row) rows per wave, lanes over the columns
            const int nrows = 16 * rows_l;
            for (int rb = wv; rb < nrows; rb += 16) {
                const float *src[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = rb + 4 * i;
                    const int c = row / rows_l, ly = row - c * rows_l;
                    const int iy = iy0 + ly;
                    ok[i] = row < nrows && c0 + c < Cin && iy >= 0 && iy < Hi;
                    src[i] = xb + ((int64_t)(ok[i] ? c0 + c : 0) * Hi + (ok[i] ? iy : 0)) * Wi;
                }
                for (int lx0 = 0; lx0 < WL; lx0 += 128) {
                    float v[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int ix = ix0 + lx0 + 64 * h + lane;
                            v[i][h] = (ok[i] && ix >= 0 && ix < Wi) ? src[i][ix] : 0.f;
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int lx = lx0 + 64 * h + lane;
                            if (rb + 4 * i < nrows && lx < WL) xs[(rb + 4 * i) * WL + lx] = v[i][h];
                        }
                }
            }
        }
        const int TP4 = (TP + 3) & ~3;
        {                                                                  // dy tile: (output channel, output row) rows per wave
            const int nrows = NB * rows;
            for (int rb = wv; rb < nrows; rb += 16) {
                const float *src[4];
                float *dst[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = rb + 4 * i;
                    const int n = row / rows, oy = row - n * rows;
                    ok[i] = row < nrows && n0 + n < Cout;
                    src[i] = dyb + ((int64_t)(ok[i] ? n0 + n : 0) * Ho + r0 + (row < nrows ? oy : 0)) * Wo + x0;
                    dst[i] = ds + n * DP + oy * cols;
                }
                for (int lx0 = 0; lx0 < cols; lx0 += 128) {
                    float v[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int ox = lx0 + 64 * h + lane;
                            v[i][h] = (ok[i] && ox < cols) ? src[i][ox] : 0.f;
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int ox = lx0 + 64 * h + lane;
                            if (rb + 4 * i < nrows && ox < cols) dst[i][ox] = v[i][h];
                        }
                }
            }
        }
        if (TP4 > TP)
            for (int i = tid; i < NB * (TP4 - TP); i += kF32Threads) ds[(i / (TP4 - TP)) * DP + TP + i % (TP4 - TP)] = 0.f;
        __syncthreads();
        for (int k0 = 4 * pw; k0 < TP4; k0 += 4 * PW) {
            const float a = ds[(nw * 16 + i16) * DP + k0 + g];             // A[n = i16][k = pixel k0 + g]
            int q = k0 + g;                                               // B[k = pixel][c = i16]
            if (q >= TP) q = 0;                                           // its dy is zero
            const int oy = q / cols, ox = q - oy * cols;
            const float *xr = xs + i16 * xelems + (oy * S) * WL + ox * S;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[(tap / KS) * WL + (tap % KS)], acc[tap], 0, 0, 0);
        }
    }
    // D lane l reg r: n = 4 g + r (+ 16 nw), c = i16
    const int c = c0 + i16;
    if (c < CP16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + nw * 16 + 4 * g + r;
            if (n < NP16) {
                float *dst = part + ((((int64_t)split * PW + pw) * NP16 + n) * CP16 + c) * TAPS;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) dst[t] = acc[t][r];
            }
        }
    }
}

// zero-insertion upsampling: out[b, c, 2 y + oy0, 2 x + ox0] = in[b, c, y, x], zeros elsewhere (data gradient of a stride-2 conv)
__global__ void upsample2_zero_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t planes, int H, int W, int Ho, int Wo) {
    const int64_t total = planes * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const int64_t p = i / ((int64_t)Wo * Ho);
        float v = 0.f;
        if (!(oy & 1) && !(ox & 1) && (oy >> 1) < H && (ox >> 1) < W) v = in[(p * H + (oy >> 1)) * W + (ox >> 1)];
        out[i] = v;
    }
}

static void f32_plan(int Ho, int Wo, int *R, int *CW, int *strips, int *ctiles) {
    *CW = Wo <= 160 ? Wo : 160;
    while (Wo > 160 && Wo % *CW) --*CW;                   // equal column tiles where the width allows (320 -> 160)
    *R = 160 / *CW < 1 ? 1 : 160 / *CW;
    if (*R > Ho) *R = Ho;
    *strips = (Ho + *R - 1) / *R;
    *ctiles = (Wo + *CW - 1) / *CW;
}

// Forward plan: NWN waves along the output channels (1 / 2 / 4 for <= 16 / <= 32 / more channels), strips of R output rows x CW
// columns with at most 160 * (4 / NWN) pixels (10 MFMA tiles per wave), the strip halved while the launch has fewer than ~3
// workgroups per CU (small maps: 16 x 20 x 20 pixels made 48 workgroups of 160 pixels) and while the LDS tile exceeds half a CU's.
static void f32_fwd_plan(int B, int Cin, int Cout, int Ho, int Wo, int KS, int S, int *NWN, int *R, int *CW, int *strips, int *ctiles, int *kc,
                         size_t *ldsb) {
    const int nb16 = (Cout + 15) / 16;
    *NWN = nb16 >= 3 ? 4 : nb16;
    *kc = Cin >= 16 ? 16 : (Cin + 3) / 4 * 4;
    const int cap = 160 * (4 / *NWN);
    *CW = Wo <= 160 ? Wo : 160;
    while (Wo > 160 && Wo % *CW) --*CW;                   // equal column tiles where the width allows (320 -> 160)
    *ctiles = (Wo + *CW - 1) / *CW;
    int r = cap / *CW < 1 ? 1 : cap / *CW;
    if (r > Ho) r = Ho;
    const int nblk = (Cout + 16 * *NWN - 1) / (16 * *NWN);
    auto lds = [&](int rr) {
        const int rows_l = (rr - 1) * S + KS, WL = (*CW - 1) * S + KS;
        return sizeof(float) * ((((size_t)*kc * rows_l * WL + 3) & ~(size_t)3) + (size_t)KS * KS * 16 * *NWN * kF32WP);
    };
    while (r > 1 && lds(r) > 80 * 1024) --r;
    // strip height by a small cost model (measured constants): a launch runs in rounds of 512 workgroups (two per CU), a stage of
    // a workgroup costs ~3 us of staging latency + 0.48 us per pixel tile of its MFMA section size (2 / 5 / 10 tiles per wave),
    // twice that when two workgroups share the SIMDs.  Candidates: the largest strip and its halvings; ties go to the smaller.
    auto cost = [&](int rr) {
        const int64_t wgs = (int64_t)B * ((Ho + rr - 1) / rr) * *ctiles * nblk;
        const int tw = (((rr * *CW + 15) / 16) + 4 / *NWN - 1) / (4 / *NWN);       // tiles per wave
        const int sz = tw <= 2 ? 2 : tw <= 5 ? 5 : 10;
        return (double)((wgs + 511) / 512) * (3.0 + 0.48 * sz * (wgs > 256 ? 2.0 : 1.0));
    };
    int best = r;
    for (int rr = r; rr > 1;) {
        rr = (rr + 1) / 2;
        if (cost(rr) <= cost(best)) best = rr;
    }
    r = best;
    *R = r;
    *strips = (Ho + r - 1) / r;
    *ldsb = lds(r);
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_conv_f32_packed_elems(int Cout, int Cin, int KS, int dgrad) {
    const int n = dgrad ? Cin : Cout, k = dgrad ? Cout : Cin;
    return (int64_t)KS * KS * ((n + 63) / 64 * 64) * ((k + 15) / 16 * 16);
}

int dfine_conv_f32_pack_weights(const float *w, float *w2, int Cout, int Cin, int KS, int dgrad, void *stream) {
    if (!w || !w2 || Cout < 1 || Cin < 1 || KS < 1 || KS > 3) return DFINE_E_BADARG;
    const int n = dgrad ? Cin : Cout, k = dgrad ? Cout : Cin;
    const int NP = (n + 63) / 64 * 64, KP = (k + 15) / 16 * 16;
    const int64_t total = (int64_t)KS * KS * NP * KP;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_pack_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, w2, Cout, Cin, KS, NP, KP, dgrad);
    return check_launch();
}

// y [B, Cout, Ho, Wo] = conv(x [B, Cin, Hi, Wi], w2) in exact fp32: kernel size KS in {1, 2, 3}, stride S in {1, 2}, padding
// (pt, pl) on the top / left (the bottom / right padding follows from Ho, Wo: anything read outside the input is zero).
int dfine_conv_f32_fwd(const float *x, const float *w2, float *y, int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int KS, int S, int pt,
                       int pl, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !w2 || !y || Cin < 1 || Cout < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || KS < 1 || KS > 3 || (S != 1 && S != 2) || pt < 0 || pl < 0)
        return DFINE_E_BADARG;
    int NWN, R, CW, strips, ctiles, kc;
    size_t ldsb;
    f32_fwd_plan(B, Cin, Cout, Ho, Wo, KS, S, &NWN, &R, &CW, &strips, &ctiles, &kc, &ldsb);
    const int NP = (Cout + 63) / 64 * 64, KP = (Cin + 15) / 16 * 16;
    if (ldsb > 160 * 1024) return DFINE_E_BADARG;
    const dim3 grid(B * strips * ctiles, (Cout + 16 * NWN - 1) / (16 * NWN));
    hipStream_t st = (hipStream_t)stream;
#define DFINE_F32_CONV_N(KSS, SS, NW)                                                                                                 \
    {                                                                                                                                 \
        static bool attr = false;                                                                                                     \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_f32_kernel<KSS, SS, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((conv_f32_kernel<KSS, SS, NW>), grid, dim3(kF32Threads), ldsb, st, x, w2, y, Cin, Cout, NP, KP, Hi, Wi, Ho, Wo, pt, pl, R, CW, strips, ctiles, kc); \
    }
#define DFINE_F32_CONV(KSS, SS)                                                                                                       \
    {                                                                                                                                 \
        if (NWN == 1) DFINE_F32_CONV_N(KSS, SS, 1)                                                                                    \
        else if (NWN == 2) DFINE_F32_CONV_N(KSS, SS, 2)                                                                               \
        else DFINE_F32_CONV_N(KSS, SS, 4)                                                                                             \
    }
    if (KS == 1 && S == 1) DFINE_F32_CONV(1, 1)
    else if (KS == 1 && S == 2) DFINE_F32_CONV(1, 2)
    else if (KS == 2 && S == 1) DFINE_F32_CONV(2, 1)
    else if (KS == 2 && S == 2) DFINE_F32_CONV(2, 2)
    else if (KS == 3 && S == 1) DFINE_F32_CONV(3, 1)
    else DFINE_F32_CONV(3, 2)
#undef DFINE_F32_CONV
#undef DFINE_F32_CONV_N
    return check_launch();
}

static void f32_wgrad_plan(int B, int Cin, int Cout, int Ho, int Wo, int KS, int *NWN, int *R, int *CW, int *strips, int *ctiles, int *splits,
                           int *ups) {
    f32_plan(Ho, Wo, R, CW, strips, ctiles);
    const int nb16 = (Cout + 15) / 16;
    *NWN = nb16 >= 3 ? 4 : nb16;
    const int pw = 4 / *NWN;                              // slabs per block
    const int units = B * *strips * *ctiles;
    const int tiles = ((Cout + 16 * *NWN - 1) / (16 * *NWN)) * ((Cin + 15) / 16);
    int sp = 2048 / tiles;
    if (pw > 1) sp = sp / pw > 768 / tiles ? sp / pw : (768 / tiles < sp ? 768 / tiles : sp);
    if (sp < 1) sp = 1;
    const int64_t bytes_per_slab = (int64_t)((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * KS * KS * 4;
    int cap = (int)(32000000 / (bytes_per_slab * pw));
    if (cap < 4) cap = 4;
    if (sp > cap) sp = cap;
    if (sp > units) sp = units;
    *ups = (units + sp - 1) / sp;
    *splits = (units + *ups - 1) / *ups;
}

// number of partial slabs dfine_conv_f32_wgrad writes
int dfine_conv_f32_wgrad_splits(int B, int Cin, int Cout, int Ho, int Wo, int KS) {
    int NWN, R, CW, strips, ctiles, splits, ups;
    f32_wgrad_plan(B, Cin, Cout, Ho, Wo, KS, &NWN, &R, &CW, &strips, &ctiles, &splits, &ups);
    return splits * (4 / NWN);
}

// part [slabs][NP16][CP16][KS*KS] f32 (NP16 / CP16 = Cout / Cin rounded up to 16, slabs = dfine_conv_f32_wgrad_splits): partial
// sums of dW = sum_{b, p} dy x, to be reduced by dfine_multi_wgrad_reduce / summed by the caller.
int dfine_conv_f32_wgrad(const float *x, const float *dy, float *part, int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int KS, int S,
                         int pt, int pl, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !dy || !part || Cin < 1 || Cout < 1 || KS < 1 || KS > 3 || (S != 1 && S != 2)) return DFINE_E_BADARG;
    int NWN, R, CW, strips, ctiles, splits, ups;
    f32_wgrad_plan(B, Cin, Cout, Ho, Wo, KS, &NWN, &R, &CW, &strips, &ctiles, &splits, &ups);
    const int np16 = (Cout + 15) / 16 * 16, cp16 = (Cin + 15) / 16 * 16;
    const int nnt = (Cout + 16 * NWN - 1) / (16 * NWN), nct = (Cin + 15) / 16;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS;
    const size_t ldsb = sizeof(float) * ((size_t)16 * rows_l * WL + (size_t)16 * NWN * (R * CW + 4));
    if (ldsb > 160 * 1024) return DFINE_E_BADARG;
    const dim3 grid(nnt * nct, splits);
    hipStream_t st = (hipStream_t)stream;
#define DFINE_F32_WG_N(KSS, SS, NW)                                                                                                   \
    {                                                                                                                                 \
        static bool attr = false;                                                                                                     \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_f32_kernel<KSS, SS, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((wgrad_f32_kernel<KSS, SS, NW>), grid, dim3(kF32Threads), ldsb, st, x, dy, part, Cin, Cout, Hi, Wi, Ho, Wo, pt, pl, R, CW, strips, ctiles, \
                           B * strips * ctiles, ups, nct, np16, cp16);                                                                \
    }
#define DFINE_F32_WG(KSS, SS)                                                                                                         \
    {                                                                                                                                 \
        if (NWN == 1) DFINE_F32_WG_N(KSS, SS, 1)                                                                                      \
        else if (NWN == 2) DFINE_F32_WG_N(KSS, SS, 2)                                                                                 \
        else DFINE_F32_WG_N(KSS, SS, 4)                                                                                               \
    }
    if (KS == 1 && S == 1) DFINE_F32_WG(1, 1)
    else if (KS == 1 && S == 2) DFINE_F32_WG(1, 2)
    else if (KS == 2 && S == 1) DFINE_F32_WG(2, 1)
    else if (KS == 2 && S == 2) DFINE_F32_WG(2, 2)
    else if (KS == 3 && S == 1) DFINE_F32_WG(3, 1)
    else DFINE_F32_WG(3, 2)
#undef DFINE_F32_WG
#undef DFINE_F32_WG_N
    return check_launch();
}

// out [planes, 2 H + eh, 2 W + ew] = in [planes, H, W] with zeros inserted between the pixels (eh / ew in {0, 1}: the input
// extent a stride-2 convolution with an odd remainder covered)
int dfine_upsample2_zero_f32(const float *in, float *out, int64_t planes, int H, int W, int Ho, int Wo, void *stream) {
    if (planes == 0) return DFINE_OK;
    if (!in || !out || H < 1 || W < 1 || Ho < 2 * H - 1 || Wo < 2 * W - 1) return DFINE_E_BADARG;
    const int64_t total = planes * Ho * Wo;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upsample2_zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, planes, H, W, Ho, Wo);
    return check_launch();
}

}  // extern "C"
