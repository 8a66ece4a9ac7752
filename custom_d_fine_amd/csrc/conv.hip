// A1/A2 - dense 1x1 / 3x3 (stride 1, "same" padding) convolution as an implicit GEMM on the MFMA
// units, NCHW bf16, fp32 accumulate.  Used for the forward pass and - with the weights repacked
// (channels swapped, taps flipped) - for the data gradient.
//
// Reference call sites: nn.Conv2d inside ConvBNAct / ConvNormLayer(_fuse) / VGGBlock
// (src/d_fine/arch/hgnetv2.py:35-80, src/d_fine/arch/hybrid_encoder.py:21-156).  ATen hands them to
// MIOpen, which on gfx950 spends 36 ms per D-FINE-m step on them, most layers latency-bound
// (profiles/r01_conv_survey_miopen.txt) plus NCHW<->NHWC transposes around its NHWC igemm kernels.
//
// GEMM view per image:  Y[n, p] = sum_{tap, c} W2[tap][n][c] * X[c, p + shift(tap)]
//   M = output channels (A operand = packed weights, c contiguous -> k-packed fragments straight
//       from global / L2), N = pixels, K = input channels x taps.
// NCHW keeps PIXELS contiguous, but an MFMA B fragment needs 8 consecutive K (= channels) per
// pixel, so each 32-channel slab of the input strip (+1-pixel halo) is staged once in LDS
// TRANSPOSED to [pixel][32 channels] (64 B per pixel: every ds_read_b128 of a wave is one linear
// 1 KiB run, conflict-free) and then reused by all KS*KS taps (a tap is just an LDS address offset)
// and by the block's 4 waves.  Almost every layer of this network is HBM-bound at bf16 (DESIGN.md
// section 5), so the kernel is organised around reading X once per output-channel block and
// writing Y once, not around peak MFMA rate.
//   block  = 256 threads, one (image, strip of R rows, 64*NTN output channels); wave w owns
//            16*NTN output channels x the whole strip (<= 160 pixels = 10 MFMA column tiles).
//   MFMA   = v_mfma_f32_16x16x32_bf16: A lane l -> W2[n = l&15][c = 8*(l>>4)..+7],
//            B lane l -> X_lds[pixel = l&15][c = 8*(l>>4)..+7], D lane l -> Y[n = 4*(l>>4)+reg][pixel = l&15].
#include "common.h"

namespace dfine {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4v;

constexpr int kConvThreads = 256;
constexpr int kMaxPixTiles = 10;     // 160 pixels per strip

// ---- weight packing: fp32 master [Cout][Cin][KS][KS] -> bf16 [KS*KS][NP][KP] (zero padded) -----
// dgrad = 0: n = cout, k = cin.   dgrad = 1: n = cin, k = cout, taps flipped.
__global__ void conv_pack_weights_kernel(const float *__restrict__ w, uint16_t *__restrict__ w2, int Cout,
                                         int Cin, int KS, int NP, int KP, int dgrad) {
    const int64_t total = (int64_t)KS * KS * NP * KP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KP);
        const int n = (int)((i / KP) % NP);
        const int tap = (int)(i / ((int64_t)KP * NP));
        float v = 0.f;
        if (!dgrad) {
            if (n < Cout && k < Cin) v = w[((int64_t)n * Cin + k) * KS * KS + tap];
        } else {
            const int r = tap / KS, s = tap % KS;
            const int src_tap = (KS - 1 - r) * KS + (KS - 1 - s);
            if (n < Cin && k < Cout) v = w[((int64_t)k * Cin + n) * KS * KS + src_tap];
        }
        w2[i] = f32_to_bf16(v);
    }
}

template <int VEC> struct PixVec;
template <> struct PixVec<8> { typedef uint4 type; };
template <> struct PixVec<4> { typedef uint2 type; };
template <> struct PixVec<2> { typedef uint32_t type; };

template <int VEC> __device__ __forceinline__ void unpack(const typename PixVec<VEC>::type &v, uint16_t (&o)[VEC]);
template <> __device__ __forceinline__ void unpack<8>(const uint4 &v, uint16_t (&o)[8]) {
    o[0] = v.x & 0xffff; o[1] = v.x >> 16; o[2] = v.y & 0xffff; o[3] = v.y >> 16;
    o[4] = v.z & 0xffff; o[5] = v.z >> 16; o[6] = v.w & 0xffff; o[7] = v.w >> 16;
}
template <> __device__ __forceinline__ void unpack<4>(const uint2 &v, uint16_t (&o)[4]) {
    o[0] = v.x & 0xffff; o[1] = v.x >> 16; o[2] = v.y & 0xffff; o[3] = v.y >> 16;
}
template <> __device__ __forceinline__ void unpack<2>(const uint32_t &v, uint16_t (&o)[2]) {
    o[0] = v & 0xffff; o[1] = v >> 16;
}

// x [B, Cin, H, W] bf16; w2 [KS*KS][NP][KP] bf16 (NP = Cout rounded up to 16, KP = Cin rounded up to 32);
// y [B, Cout, H, W] bf16.  H*W, W describe the (possibly flattened, for 1x1) plane.
template <int KS, int NTN, int VEC>
__global__ __launch_bounds__(kConvThreads) void conv_igemm_kernel(
    const uint16_t *__restrict__ x, const uint16_t *__restrict__ w2, uint16_t *__restrict__ y, int Cin,
    int Cout, int NP, int KP, int H, int W, int R, int strips) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PAD = KS / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / strips, strip = blockIdx.x - b * strips;
    const int r0 = strip * R;
    const int rows = min(R, H - r0);
    const int TP = rows * W;                           // valid output pixels of this strip
    const int WL = W + 2 * PAD, rows_l = R + 2 * PAD;
    const int npxl = rows_l * WL;
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(lds);
    for (int i = tid; i < npxl * 16; i += kConvThreads) lds32[i] = 0u;

    // per column-tile LDS byte offset of this lane's pixel (tap (0,0)) + its 16-byte channel group
    int pl[kMaxPixTiles];
    const int ntile = (TP + 15) / 16;
#pragma unroll
    for (int jt = 0; jt < kMaxPixTiles; ++jt) {
        int q = jt * 16 + (lane & 15);
        if (q >= TP) q = 0;
        const int orow = q / W, ocol = q - orow * W;
        pl[jt] = (orow * WL + ocol) * 64 + (lane >> 4) * 16;
    }
    const int n_wave = blockIdx.y * 64 * NTN + wave * 16 * NTN;
    f32x4v acc[NTN][kMaxPixTiles];
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
        for (int jt = 0; jt < kMaxPixTiles; ++jt) acc[t][jt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int nvec_row = W / VEC;
    const int nvec = rows_l * nvec_row;
    const uint16_t *xb = x + (int64_t)b * Cin * H * W;
    __syncthreads();

    for (int c0 = 0; c0 < KP; c0 += 32) {
        // ---- stage the [32 channels] x [strip + halo] slab, transposed to [pixel][channel] -----
        for (int it = tid; it < 16 * nvec; it += kConvThreads) {
            const int pair = it & 15, v = it >> 4;
            const int lr = v / nvec_row, xv = (v - lr * nvec_row) * VEC;
            const int gy = r0 - PAD + lr;
            if (gy < 0 || gy >= H) continue;                   // stays zero (never written)
            const int ca = c0 + 2 * pair;
            uint16_t e0[VEC], e1[VEC];
            if (ca < Cin) {
                unpack<VEC>(*reinterpret_cast<const typename PixVec<VEC>::type *>(xb + ((int64_t)ca * H + gy) * W + xv), e0);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) e0[i] = 0;
            }
            if (ca + 1 < Cin) {
                unpack<VEC>(*reinterpret_cast<const typename PixVec<VEC>::type *>(xb + ((int64_t)(ca + 1) * H + gy) * W + xv), e1);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) e1[i] = 0;
            }
            uint32_t *dst = lds32 + (lr * WL + xv + PAD) * 16 + pair;
#pragma unroll
            for (int i = 0; i < VEC; ++i) dst[i * 16] = (uint32_t)e0[i] | ((uint32_t)e1[i] << 16);
        }
        __syncthreads();
        // ---- MFMA over the taps ----------------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int toff = ((tap / KS) * WL + (tap % KS)) * 64;
            bf16x8 a[NTN];
#pragma unroll
            for (int t = 0; t < NTN; ++t) {
                const int n = n_wave + t * 16 + (lane & 15);
                uint4 av = make_uint4(0, 0, 0, 0);
                if (n < NP) av = *reinterpret_cast<const uint4 *>(w2 + ((int64_t)tap * NP + n) * KP + c0 + 8 * (lane >> 4));
                a[t] = __builtin_bit_cast(bf16x8, av);
            }
#pragma unroll
            for (int jt = 0; jt < kMaxPixTiles; ++jt) {
                if (jt < ntile) {
                    const uint4 bv = *reinterpret_cast<const uint4 *>(lds + pl[jt] + toff);
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
                    for (int t = 0; t < NTN; ++t)
                        acc[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bf, acc[t][jt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- store: lane holds 4 consecutive output channels for one pixel of every column tile -----
    uint16_t *yb = y + ((int64_t)b * Cout * H + r0) * W;
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
#pragma unroll
        for (int jt = 0; jt < kMaxPixTiles; ++jt) {
            if (jt < ntile) {
                const int q = jt * 16 + (lane & 15);
                if (q < TP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n_wave + t * 16 + 4 * (lane >> 4) + r;
                        if (n < Cout) yb[(int64_t)n * H * W + q] = f32_to_bf16(acc[t][jt][r]);
                    }
                }
            }
        }
    }
}

static int launch_conv(const uint16_t *x, const uint16_t *w2, uint16_t *y, int B, int Cin, int Cout, int NP, int KP,
                       int H, int W, int KS, hipStream_t st) {
    // strip height: as many rows as fit in 160 pixels
    int R = 160 / W;
    if (R < 1) return DFINE_E_BADARG;
    if (R > H) R = H;
    const int strips = (H + R - 1) / R;
    const int pad = KS / 2;
    const size_t ldsb = (size_t)(R + 2 * pad) * (W + 2 * pad) * 64;
    const int vec = (W % 8 == 0) ? 8 : (W % 4 == 0 ? 4 : 2);
    const int nblk64 = (NP + 63) / 64;
    const bool wide = (NP % 128 == 0) && ((int64_t)B * strips * (NP / 128) >= 512);
    dim3 grid(B * strips, wide ? NP / 128 : nblk64);
#define DFINE_CONV(KSS, NTNN, VECC)                                                                   \
    hipLaunchKernelGGL((conv_igemm_kernel<KSS, NTNN, VECC>), grid, dim3(kConvThreads), ldsb, st, x, w2, y, Cin, \
                       Cout, NP, KP, H, W, R, strips)
#define DFINE_CONV_V(KSS, NTNN)                                                                       \
    { if (vec == 8) DFINE_CONV(KSS, NTNN, 8); else if (vec == 4) DFINE_CONV(KSS, NTNN, 4); else DFINE_CONV(KSS, NTNN, 2); }
    if (KS == 3) { if (wide) DFINE_CONV_V(3, 2) else DFINE_CONV_V(3, 1) }
    else if (KS == 1) { if (wide) DFINE_CONV_V(1, 2) else DFINE_CONV_V(1, 1) }
    else return DFINE_E_BADARG;
#undef DFINE_CONV_V
#undef DFINE_CONV
    return check_launch();
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int64_t dfine_conv_packed_elems(int Cout, int Cin, int KS, int dgrad) {
    const int n = dgrad ? Cin : Cout, k = dgrad ? Cout : Cin;
    return (int64_t)KS * KS * ((n + 15) / 16 * 16) * ((k + 31) / 32 * 32);
}

int dfine_conv_pack_weights(const float *w, void *w2, int Cout, int Cin, int KS, int dgrad, void *stream) {
    if (!w || !w2 || Cout < 1 || Cin < 1 || (KS != 1 && KS != 3)) return DFINE_E_BADARG;
    const int n = dgrad ? Cin : Cout, k = dgrad ? Cout : Cin;
    const int NP = (n + 15) / 16 * 16, KP = (k + 31) / 32 * 32;
    const int64_t total = (int64_t)KS * KS * NP * KP;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (uint16_t *)w2, Cout,
                       Cin, KS, NP, KP, dgrad);
    return check_launch();
}

// y[B, Cout, H, W] = conv(x[B, Cin, H, W], packed weights), stride 1, padding KS/2, bf16.
// `w2` comes from dfine_conv_pack_weights(dgrad = 0) - or (dgrad = 1) with Cin/Cout exchanged by the
// caller, which makes this the data gradient dX = conv(dY, flipped-transposed weights).
int dfine_conv_fwd_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int H, int W, int KS,
                        void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !w2 || !y || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (KS != 1 && KS != 3)) return DFINE_E_BADARG;
    if (Cin % 2) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    int h = H, w = W;
    if (KS == 1) {             // no spatial structure: treat the plane as rows of <= 160 pixels
        const int hw = H * W;
        w = 160;
        while (w > 1 && (hw % w || w % 2)) --w;
        if (w < 16) { h = H; w = W; } else h = hw / w;
    }
    if (w % 2 || w > 160) return DFINE_E_BADARG;
    return launch_conv((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, h, w, KS,
                       (hipStream_t)stream);
}

}  // extern "C"
