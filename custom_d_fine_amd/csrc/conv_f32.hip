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

namespace dfine {

typedef __attribute__((ext_vector_type(4))) float f4v;
constexpr int kF32Threads = 256, kF32KC = 16, kF32MaxTiles = 10;

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

template <int KS, int S>
__global__ __launch_bounds__(kF32Threads) void conv_f32_kernel(const float *__restrict__ x, const float *__restrict__ w2, float *__restrict__ y,
                                                               int Cin, int Cout, int NP, int KP, int Hi, int Wi, int Ho, int Wo, int pt, int pl,
                                                               int R, int CW, int strips, int ctiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int TAPS = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int g = lane >> 4, i16 = lane & 15;
    int u = blockIdx.x;
    const int ct = u % ctiles; u /= ctiles;
    const int strip = u % strips; const int b = u / strips;
    const int r0 = strip * R, x0 = ct * CW;
    const int rows = min(R, Ho - r0), cols = min(CW, Wo - x0);
    const int TP = rows * cols;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS;
    const int iy0 = r0 * S - pt, ix0 = x0 * S - pl;
    float *xs = reinterpret_cast<float *>(lds_raw);                    // [kF32KC][rows_l][WL]
    float *ws = xs + kF32KC * rows_l * WL;                             // [TAPS][64][kF32KC + 1]
    constexpr int WP = kF32KC + 1;
    const int n0 = blockIdx.y * 64;
    const int ntile = (TP + 15) / 16;
    // LDS offset of this lane's pixel (tap 0) per pixel tile
    int pl_off[kF32MaxTiles];
#pragma unroll
    for (int jt = 0; jt < kF32MaxTiles; ++jt) {
        int q = jt * 16 + i16;
        if (q >= TP) q = 0;
        const int oy = q / cols, ox = q - oy * cols;
        pl_off[jt] = (oy * S) * WL + ox * S;
    }
    f4v acc[kF32MaxTiles];
#pragma unroll
    for (int jt = 0; jt < kF32MaxTiles; ++jt) acc[jt] = f4v{0.f, 0.f, 0.f, 0.f};
    const float *xb = x + (int64_t)b * Cin * Hi * Wi;
    const int xelems = rows_l * WL;
    for (int c0 = 0; c0 < KP; c0 += kF32KC) {
        __syncthreads();
        // one wave per (channel, tile row): the row / channel split runs on the scalar unit, lanes walk the columns (an integer
        // division pair per ELEMENT made this loop the kernel's bottleneck on the 3-channel stem: 935 us for 1.4 GFLOP)
        const int kkn = (min(kF32KC, Cin - c0) + 3) >> 2;                  // 4-channel groups of this stage that hold data
        for (int row = wv; row < 4 * kkn * rows_l; row += 4) {
            const int c = row / rows_l, ly = row - c * rows_l;
            const int iy = iy0 + ly;
            const bool ok = c0 + c < Cin && iy >= 0 && iy < Hi;
            const float *src = xb + ((int64_t)(c0 + c) * Hi + (ok ? iy : 0)) * Wi;
            float *dst = xs + row * WL;
            for (int lx = lane; lx < WL; lx += 64) {
                const int ix = ix0 + lx;
                dst[lx] = (ok && ix >= 0 && ix < Wi) ? src[ix] : 0.f;
            }
        }
        for (int i = tid; i < TAPS * 64 * kF32KC; i += kF32Threads) {
            const int k = i % kF32KC, n = (i / kF32KC) % 64, tap = i / (kF32KC * 64);
            ws[(tap * 64 + n) * WP + k] = w2[((int64_t)tap * NP + n0 + n) * KP + c0 + k];
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = (tap / KS) * WL + (tap % KS);
#pragma unroll
            for (int kk = 0; kk < kF32KC / 4; ++kk) {
                if (kk >= kkn) break;                                      // uniform: channel groups past Cin (the stem: 3 channels)
                const float a = ws[(tap * 64 + wave * 16 + i16) * WP + kk * 4 + g];
                const float *xr = xs + (kk * 4 + g) * xelems + toff;
#pragma unroll
                for (int jt = 0; jt < kF32MaxTiles; ++jt)      // tiles past the strip read pixel 0 and are never stored: no branches
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[pl_off[jt]], acc[jt], 0, 0, 0);
            }
        }
    }
    float *yb = y + (int64_t)b * Cout * Ho * Wo;
#pragma unroll
    for (int jt = 0; jt < kF32MaxTiles; ++jt) {
        const int q = jt * 16 + i16;
        if (jt < ntile && q < TP) {
            const int oy = q / cols, ox = q - oy * cols;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wave * 16 + 4 * g + r;
                if (n < Cout) yb[((int64_t)n * Ho + r0 + oy) * Wo + x0 + ox] = acc[jt][r];
            }
        }
    }
}

// part[split][NP16][CP16][TAPS]; block = (64-row n tile, 16-column c tile) x split
template <int KS, int S>
__global__ __launch_bounds__(kF32Threads) void wgrad_f32_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part,
                                                                int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int pt, int pl, int R, int CW,
                                                                int strips, int ctiles, int total_units, int units_per_split, int nct, int NP16,
                                                                int CP16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int TAPS = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int g = lane >> 4, i16 = lane & 15;
    const int nt = blockIdx.x / nct, ctile = blockIdx.x - nt * nct;
    const int n0 = nt * 64, c0 = ctile * 16;
    const int split = blockIdx.y;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS, xelems = rows_l * WL;
    const int TPmax = R * CW, DP = TPmax + 4;                              // dy tile pitch
    float *xs = reinterpret_cast<float *>(lds_raw);                       // [16 c][rows_l][WL]
    float *ds = xs + 16 * xelems;                                          // [64 n][DP]
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
        for (int row = wv; row < 16 * rows_l; row += 4) {                  // (channel, tile row) per wave, lanes over the columns
            const int c = row / rows_l, ly = row - c * rows_l;
            const int iy = iy0 + ly;
            const bool ok = c0 + c < Cin && iy >= 0 && iy < Hi;
            const float *src = xb + ((int64_t)(c0 + c) * Hi + (ok ? iy : 0)) * Wi;
            float *dst = xs + row * WL;
            for (int lx = lane; lx < WL; lx += 64) {
                const int ix = ix0 + lx;
                dst[lx] = (ok && ix >= 0 && ix < Wi) ? src[ix] : 0.f;
            }
        }
        const int TP4 = (TP + 3) & ~3;
        for (int row = wv; row < 64 * rows; row += 4) {                    // (output channel, output row) per wave
            const int n = row / rows, oy = row - n * rows;
            const bool ok = n0 + n < Cout;
            const float *src = dyb + ((int64_t)(ok ? n0 + n : 0) * Ho + r0 + oy) * Wo + x0;
            float *dst = ds + n * DP + oy * cols;
            for (int ox = lane; ox < cols; ox += 64) dst[ox] = ok ? src[ox] : 0.f;
        }
        if (TP4 > TP)
            for (int i = tid; i < 64 * (TP4 - TP); i += kF32Threads) ds[(i / (TP4 - TP)) * DP + TP + i % (TP4 - TP)] = 0.f;
        __syncthreads();
        for (int k0 = 0; k0 < TP4; k0 += 4) {
            const float a = ds[(wave * 16 + i16) * DP + k0 + g];           // A[n = i16][k = pixel k0 + g]
            int q = k0 + g;                                               // B[k = pixel][c = i16]
            if (q >= TP) q = 0;                                           // its dy is zero
            const int oy = q / cols, ox = q - oy * cols;
            const float *xr = xs + i16 * xelems + (oy * S) * WL + ox * S;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[(tap / KS) * WL + (tap % KS)], acc[tap], 0, 0, 0);
        }
    }
    // D lane l reg r: n = 4 g + r (+ 16 wave), c = i16
    const int c = c0 + i16;
    if (c < CP16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wave * 16 + 4 * g + r;
            if (n < NP16) {
                float *dst = part + (((int64_t)split * NP16 + n) * CP16 + c) * TAPS;
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
    int R, CW, strips, ctiles;
    f32_plan(Ho, Wo, &R, &CW, &strips, &ctiles);
    const int NP = (Cout + 63) / 64 * 64, KP = (Cin + 15) / 16 * 16;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS;
    const size_t ldsb = sizeof(float) * ((size_t)kF32KC * rows_l * WL + (size_t)KS * KS * 64 * (kF32KC + 1));
    if (ldsb > 160 * 1024) return DFINE_E_BADARG;
    const dim3 grid(B * strips * ctiles, NP / 64);
    hipStream_t st = (hipStream_t)stream;
#define DFINE_F32_CONV(KSS, SS)                                                                                                       \
    {                                                                                                                                 \
        static bool attr = false;                                                                                                     \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_f32_kernel<KSS, SS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((conv_f32_kernel<KSS, SS>), grid, dim3(kF32Threads), ldsb, st, x, w2, y, Cin, Cout, NP, KP, Hi, Wi, Ho, Wo, pt, pl, R, CW, strips, ctiles); \
    }
    if (KS == 1 && S == 1) DFINE_F32_CONV(1, 1)
    else if (KS == 1 && S == 2) DFINE_F32_CONV(1, 2)
    else if (KS == 2 && S == 1) DFINE_F32_CONV(2, 1)
    else if (KS == 2 && S == 2) DFINE_F32_CONV(2, 2)
    else if (KS == 3 && S == 1) DFINE_F32_CONV(3, 1)
    else DFINE_F32_CONV(3, 2)
#undef DFINE_F32_CONV
    return check_launch();
}

static void f32_wgrad_plan(int B, int Cin, int Cout, int Ho, int Wo, int KS, int *R, int *CW, int *strips, int *ctiles, int *splits, int *ups) {
    f32_plan(Ho, Wo, R, CW, strips, ctiles);
    const int units = B * *strips * *ctiles;
    const int tiles = ((Cout + 63) / 64) * ((Cin + 15) / 16);
    int sp = 2048 / tiles;
    if (sp < 1) sp = 1;
    const int64_t bytes_per_split = (int64_t)((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * KS * KS * 4;
    int cap = (int)(32000000 / bytes_per_split);
    if (cap < 4) cap = 4;
    if (sp > cap) sp = cap;
    if (sp > units) sp = units;
    *ups = (units + sp - 1) / sp;
    *splits = (units + *ups - 1) / *ups;
}

int dfine_conv_f32_wgrad_splits(int B, int Cin, int Cout, int Ho, int Wo, int KS) {
    int R, CW, strips, ctiles, splits, ups;
    f32_wgrad_plan(B, Cin, Cout, Ho, Wo, KS, &R, &CW, &strips, &ctiles, &splits, &ups);
    return splits;
}

// part [splits][NP16][CP16][KS*KS] f32 (NP16 / CP16 = Cout / Cin rounded up to 16): per-split partial sums of
// dW = sum_{b, p} dy x, to be reduced by dfine_multi_wgrad_reduce / summed by the caller.
int dfine_conv_f32_wgrad(const float *x, const float *dy, float *part, int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int KS, int S,
                         int pt, int pl, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !dy || !part || Cin < 1 || Cout < 1 || KS < 1 || KS > 3 || (S != 1 && S != 2)) return DFINE_E_BADARG;
    int R, CW, strips, ctiles, splits, ups;
    f32_wgrad_plan(B, Cin, Cout, Ho, Wo, KS, &R, &CW, &strips, &ctiles, &splits, &ups);
    const int np16 = (Cout + 15) / 16 * 16, cp16 = (Cin + 15) / 16 * 16;
    const int nnt = (Cout + 63) / 64, nct = (Cin + 15) / 16;
    const int rows_l = (R - 1) * S + KS, WL = (CW - 1) * S + KS;
    const size_t ldsb = sizeof(float) * ((size_t)16 * rows_l * WL + (size_t)64 * (R * CW + 4));
    if (ldsb > 160 * 1024) return DFINE_E_BADARG;
    const dim3 grid(nnt * nct, splits);
    hipStream_t st = (hipStream_t)stream;
#define DFINE_F32_WG(KSS, SS)                                                                                                         \
    {                                                                                                                                 \
        static bool attr = false;                                                                                                     \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_f32_kernel<KSS, SS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((wgrad_f32_kernel<KSS, SS>), grid, dim3(kF32Threads), ldsb, st, x, dy, part, Cin, Cout, Hi, Wi, Ho, Wo, pt, pl, R, CW, strips, ctiles, \
                           B * strips * ctiles, ups, nct, np16, cp16);                                                                \
    }
    if (KS == 1 && S == 1) DFINE_F32_WG(1, 1)
    else if (KS == 1 && S == 2) DFINE_F32_WG(1, 2)
    else if (KS == 2 && S == 1) DFINE_F32_WG(2, 1)
    else if (KS == 2 && S == 2) DFINE_F32_WG(2, 2)
    else if (KS == 3 && S == 1) DFINE_F32_WG(3, 1)
    else DFINE_F32_WG(3, 2)
#undef DFINE_F32_WG
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
