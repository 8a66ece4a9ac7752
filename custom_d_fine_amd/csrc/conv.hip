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
// NCHW keeps PIXELS contiguous, but an MFMA B fragment needs 8 consecutive K (= channels) per pixel.
//   1x1 layers (conv1x1_tr_kernel): X rows go to LDS as they are ([channel][pixel], 16-byte copies) and the
//   fragments come from gfx950's LDS transpose-read, ds_read_b64_tr_b16 - see that kernel.
//   3x3 layers (conv_igemm_kernel): a tap shifts the pixel index by +-1 = 2 bytes, which the transpose-read's
//   8-byte alignment cannot follow, so each 32-channel slab of the input strip (+1-pixel halo) is staged in LDS
// TRANSPOSED to [pixel][32 channels] (64 B per pixel) and then reused by all KS*KS taps (a tap is just an LDS
// address offset) and by the block's 4 waves.  PMC: ds_read_b128's non-contiguous 16-lane service groups make
// pixels p and p + 4 share a 16-byte slot here (2-way conflict, 49 % of the LDS cycles); the conflict-free
// [K group][pixel][8 channels] image was measured and is slower overall (the staging loop is
// VALU bound).  Almost every layer of this network is HBM-bound at bf16 (DESIGN.md
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

// csrc/wgrad3.hip: the row-streaming 3x3 weight gradient (W % 8 == 0, W <= 160); 0 splits = shape not taken
int wgrad3_rows_splits(int B, int Cin, int Cout, int H, int W);
int wgrad3_rows_launch(const void *x, const void *dy, float *part, int B, int Cin, int Cout, int H, int W, hipStream_t st);

// csrc/conv3s.hip: row-streaming 3x3 forward / data gradient for <= 32 channels on both sides
bool conv3x3_rows32_ok(int NP, int KP, int H, int W);

// Per-output-channel affine + activation applied to the fp32 accumulators in the store phase of the forward kernels:
//     y[n] = lab[0] * act(scale[n] * conv[n] + shift[n]) + lab[1]
// An eval-mode BatchNorm (scale / shift folded from the running statistics), a deployed layer's bias (scale = 1) and the
// learnable affine block behind the activation (ref hgnetv2.py:35-80, hybrid_encoder.py:21-79) - the inference forward
// then is ONE launch per conv -> BN -> act unit and the map is written once instead of written, read and written again.
// scale == nullptr: plain store.
struct EpiAffine {
    const float *scale, *shift, *lab;
    int act;
};
__device__ __forceinline__ float epi_act(float z, int act) {
    if (act == 1) return fmaxf(z, 0.f);
    if (act == 2) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));      // the expression of bnact.hip's act_fwd
    return z;
}
int conv3x3_rows32_launch(const uint16_t *x, const uint16_t *w2, uint16_t *y, int B, int Cin, int Cout, int NP, int KP, int H, int W,
                          int accum, hipStream_t st);

constexpr int kConvThreads = 256;
constexpr int kMaxPixTiles = 10;     // 160 pixels per strip

// 1x1 layers feed the MFMA B operand with ds_read_b64_tr_b16 (conv1x1_tr_kernel): one transpose-read returns, for
// the lane's pixel, 4 channels that are 4 consecutive LDS rows, and the four 16-lane groups of a wave read 16
// distinct rows per instruction.  MFMA k index 8 g + e of a 32-channel slab therefore stands for channel
// (e < 4 ? 4 g + e : 16 + 4 g + e - 4); the packed weights use the same order so the A fragment stays one 16-byte load.
__host__ __device__ __forceinline__ int tr_slab_channel(int kk) {
    const int g = kk >> 3, e = kk & 7;
    return e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
}

// ---- weight packing: fp32 master [Cout][Cin][KS][KS] -> bf16 [KS*KS][NP][KP] (zero padded) -----
// dgrad = 0: n = cout, k = cin.   dgrad = 1: n = cin, k = cout, taps flipped.
__global__ void conv_pack_weights_kernel(const float *__restrict__ w, uint16_t *__restrict__ w2, int Cout,
                                         int Cin, int KS, int NP, int KP, int dgrad) {
    const int64_t total = (int64_t)KS * KS * NP * KP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k = (int)(i % KP);
        if (KS == 1) k = (k & ~31) + tr_slab_channel(k & 31);          // position -> source channel
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

// All layers in one launch: table rows of 8 x int64 = {src ptr, dst ptr, Cout, Cin, KS, NP, KP, dgrad}; blockIdx.y =
// row, blockIdx.x strides over the row's packed elements (176 pack launches per train step otherwise).
__global__ void conv_pack_weights_multi_kernel(const int64_t *__restrict__ table) {
    const int64_t *e = table + (int64_t)blockIdx.y * 8;
    const float *w = reinterpret_cast<const float *>(e[0]);
    uint16_t *w2 = reinterpret_cast<uint16_t *>(e[1]);
    const int Cout = (int)e[2], Cin = (int)e[3], KS = (int)e[4], NP = (int)e[5], KP = (int)e[6], dgrad = (int)e[7];
    const int total = KS * KS * NP * KP;
    if (dgrad) {
        // the data-gradient packing is a TRANSPOSE (n = input channel, k = output channel): element-wise, consecutive threads read
        // addresses Cin * KS^2 floats apart - one 128-byte line per 4-byte element, 156 us for the 19.6 M weights of D-FINE-m.
        // 32 x 32 tiles through LDS: rows of the source read along the input channels, rows of the packed layout written along k.
        __shared__ float tile[32][33];
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 256 threads: 8 rows per pass
        const int tn = (NP + 31) / 32, tk = KP / 32, taps = KS * KS;        // KP is a multiple of 32
        for (int t = blockIdx.x; t < taps * tn * tk; t += gridDim.x) {
            const int tap = t / (tn * tk), r = t - tap * (tn * tk), n0 = (r / tk) * 32, k0 = (r - (r / tk) * tk) * 32;
            const int rr = tap / KS, ss = tap - rr * KS;
            const int src_tap = (KS - 1 - rr) * KS + (KS - 1 - ss);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k = k0 + ty + 8 * p, n = n0 + tx;
                tile[ty + 8 * p][tx] = (n < Cin && k < Cout) ? w[((int64_t)k * Cin + n) * taps + src_tap] : 0.f;
            }
            __syncthreads();
            const int kk = KS == 1 ? tr_slab_channel(tx) : tx;              // packed position tx holds source channel kk of the slab
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int n = n0 + ty + 8 * p;
                if (n < NP) w2[((int64_t)tap * NP + n) * KP + k0 + tx] = f32_to_bf16(tile[kk][ty + 8 * p]);
            }
            __syncthreads();
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = i % KP;
        if (KS == 1) k = (k & ~31) + tr_slab_channel(k & 31);
        const int n = (i / KP) % NP;
        const int tap = i / (KP * NP);
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
// KC = number of 32-channel slabs staged per barrier pair: layers with many input channels are otherwise
// bound by the (global load -> LDS -> barrier) latency of each 32-channel step, not by MFMA or HBM.
// tools/probe/conv3x3_ablate.hip builds timing-only variants (1: no MFMAs, 2: no LDS fragment reads, 4: no staging of x,
// 8: no weight loads).
#ifndef DFINE_CONV3X3_ABLATE
#define DFINE_CONV3X3_ABLATE 0
#endif
constexpr int kAbl3 = DFINE_CONV3X3_ABLATE;

template <int KS, int NTN, int VEC, int KC>
__global__ __launch_bounds__(kConvThreads, 2) void conv_igemm_kernel(      // two workgroups per CU: at most 256 registers
    const uint16_t *__restrict__ x, const uint16_t *__restrict__ w2, uint16_t *__restrict__ y, int Cin,
    int Cout, int NP, int KP, int H, int W, int R, int strips, int accum) {
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
    for (int i = tid; i < npxl * 16 * KC; i += kConvThreads) lds32[i] = 0u;

    // LDS record of a pixel: 32 channels = 64 bytes = four 16-byte channel groups.  A B fragment is one ds_read_b128 per lane
    // (lane = 16 * group + pixel of the tile), served in lane sets {0-3, 12-15, 20-27} ...: with the groups in order, pixels
    // p and p + 4 of a set fall on the same banks (2-way conflicts on every fragment read - half of the LDS-active cycles,
    // profiles/r02_conv3x3_pmc.txt - and this kernel keeps the LDS pipe half busy even without them).  Group g of pixel P is
    // therefore stored in slot g ^ (2 if P & 4): any 16 consecutive pixels then read conflict-free, for every tap shift.
    const int g16 = (lane >> 4) * 16;
    int pl[kMaxPixTiles];
    const int ntile = (TP + 15) / 16;
#pragma unroll
    for (int jt = 0; jt < kMaxPixTiles; ++jt) {
        int q = jt * 16 + (lane & 15);
        if (q >= TP) q = 0;
        const int orow = q / W, ocol = q - orow * W;
        pl[jt] = (orow * WL + ocol) * 64 + g16;        // byte offset of this lane's pixel (tap (0, 0)) and channel group, unswizzled
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

    // Weights: a ring of three fragment sets per wave, filled two taps ahead ACROSS slabs and stages (the sequence of (stage,
    // slab, tap) steps is known up front and does not depend on the staging).  One tap ahead left ~0.13 us of MFMAs to cover
    // an L2 round trip per tap: with nothing else in the kernel the weight loads alone took 36 us of a 99 us launch on
    // 128 -> 128 @ 80 x 80 (tools/probe/conv3x3_ablate.hip).
    auto load_a = [&](int tap, int cs, bf16x8 (&a)[NTN]) {
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            const int n = min(n_wave + t * 16 + (lane & 15), NP - 1);          // rows past the layer: a valid row, never stored
            if (kAbl3 & 8) { a[t] = __builtin_bit_cast(bf16x8, make_uint4(n, tap, cs, lane)); continue; }
            a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(w2 + ((int64_t)tap * NP + n) * KP + cs + 8 * (lane >> 4)));
        }
    };
    bf16x8 aw[3][NTN];
    load_a(0, 0, aw[0]);
    load_a(1, 0, aw[1]);
    for (int c0 = 0; c0 < KP; c0 += 32 * KC) {
        // ---- stage KC x [32 channels] x [strip + halo] slabs, each transposed to [pixel][channel] ----
        for (int it = tid; it < 16 * nvec * KC && !(kAbl3 & 4); it += kConvThreads) {
            const int pair = it & 15, vs = it >> 4;
            const int slab = KC == 1 ? 0 : vs / nvec, v = KC == 1 ? vs : vs - slab * nvec;
            const int lr = v / nvec_row, xv = (v - lr * nvec_row) * VEC;
            const int gy = r0 - PAD + lr;
            if (gy < 0 || gy >= H) continue;                   // stays zero (never written)
            const int ca = c0 + 32 * slab + 2 * pair;
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
            const int P = lr * WL + xv + PAD;
            uint32_t *dst = lds32 + slab * npxl * 16 + P * 16;
#pragma unroll
            for (int i = 0; i < VEC; ++i) dst[i * 16 + (pair ^ (((P + i) & 4) << 1))] = (uint32_t)e0[i] | ((uint32_t)e1[i] << 16);
        }
        __syncthreads();
        // ---- MFMA over the slabs and taps --------------------------------------------------------
        // All pixel-tile fragments of a tap are read from LDS before its first MFMA (and its weights were requested two taps
        // earlier): the loop this replaces went load A -> per pixel tile { ds_read ->
        // s_waitcnt vmcnt(0) lgkmcnt(0) -> NTN MFMAs }, i.e. one exposed L2 round trip per tap and one exposed LDS round trip
        // per NTN MFMAs.  Pixel tiles past the strip (the last strip of a 20 x 20 map) read pixel 0 and are never stored.
#pragma unroll
        for (int slab = 0; slab < KC; ++slab) {
            const int cs = c0 + 32 * slab;
            if (KC > 1 && cs >= KP) break;
            const unsigned char *slds = lds + slab * npxl * 64;
            // (the fragment addresses of all 9 taps x 10 tiles do not depend on the slab: left alone the compiler computes the
            // 90 of them once, ahead of the loop, and spills)
#pragma unroll
            for (int jt = 0; jt < kMaxPixTiles; ++jt) asm volatile("" : "+v"(pl[jt]));
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                const int toff = ((tap / KS) * WL + (tap % KS)) * 64;
                static_assert((KS * KS) % 3 == 0, "ring positions repeat per slab");
                bf16x8 (&a)[NTN] = aw[tap % 3];
                if (tap + 2 < KS * KS) load_a(tap + 2, cs, aw[(tap + 2) % 3]);
                else if (cs + 32 < KP) load_a(tap + 2 - KS * KS, cs + 32, aw[(tap + 2) % 3]);
                bf16x8 bf[kMaxPixTiles];
#pragma unroll
                for (int jt = 0; jt < kMaxPixTiles; ++jt)
                {
                    const int A = pl[jt] + toff;           // bit 8 = bit 2 of the pixel index -> flips bit 5 (slot g ^ 2)
                    if (kAbl3 & 2) { bf[jt] = __builtin_bit_cast(bf16x8, make_uint4(A, lane, tap, jt)); continue; }
                    bf[jt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(slds + (A ^ ((A >> 3) & 32))));
                }
                // (pins the requests above the MFMAs: at this register pressure the scheduler otherwise sinks each weight load to
                // just before its use and re-serialises the fragment reads, ds_read -> wait -> 2 MFMAs)
                __builtin_amdgcn_sched_barrier(0);
                if (kAbl3 & 1) {
#pragma unroll
                    for (int jt = 0; jt < kMaxPixTiles; ++jt) asm volatile("" ::"v"(bf[jt]));
#pragma unroll
                    for (int t = 0; t < NTN; ++t) asm volatile("" ::"v"(a[t]));
                } else {
#pragma unroll
                    for (int jt = 0; jt < kMaxPixTiles; ++jt)
#pragma unroll
                        for (int t = 0; t < NTN; ++t)
                            acc[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bf[jt], acc[t][jt], 0, 0, 0);
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
                        if (n < Cout) {                           // accum: y += conv(x), bf16 + bf16 in fp32, one rounding (like a separate add)
                            uint16_t *o = yb + (int64_t)n * H * W + q;
                            const uint16_t c = f32_to_bf16(acc[t][jt][r]);       // the convolution's own bf16 result first, as a separate add would see it
                            *o = accum ? f32_to_bf16(bf16_to_f32(c) + bf16_to_f32(*o)) : c;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 forward / data gradient, second generation: wave-specialised and persistent.
// conv_igemm_kernel alternates "stage x into LDS" and "MFMA over the staged slabs" inside a workgroup and relies on the CU's
// second workgroup for overlap - but two workgroups launched together run the same phases at the same time, and the parts of
// a launch simply add up (tools/probe/conv3x3_ablate.hip on 128 -> 128 @ 80 x 80: empty kernel 23 us, + staging 25, + weight
// loads 23, + MFMAs 24 (= the matrix peak), + fragment reads 3 -> 98 us).  Here a workgroup is 4 MFMA waves + 4 loader waves
// (wave w and w + 4 share a SIMD), two LDS buffers, one barrier per stage: the loaders fill buffer (g + 1) & 1 - global rows
// -> registers -> [pixel][channel] records - while the MFMA waves work on buffer g & 1.  A workgroup walks a list of
// (image, strip, output-channel block) units, so the loaders are already on the next unit's first stage while the MFMA waves
// store the finished strip, and zero-filling / address set-up happen once per workgroup instead of once per strip.
constexpr int kWsThreads = 512;
// LDS row pitch in pixels: a multiple of 8, so that the slot swizzle (bit 2 of the pixel index) of a tap depends on its column
// offset only and the loaders' eight writes of a vector have compile-time offsets
__host__ __device__ static inline int ws_pitch(int W) { return (W + 2 + 7) & ~7; }

// Address arithmetic is kept out of the inner loops on both sides: the first version of this kernel spent 5.4 vector
// instructions per MFMA (rocprofv3: SQ_INSTS_VALU 23.5 M against 3.7 M MFMAs, VALU busy 38 us per SIMD next to 24 us of MFMA
// in a 76 us launch - per fragment read an add, the swizzle's shift / and / xor and the slab base; per loader item three
// integer divisions, 24 shift / and / or to pair the channels and 8 swizzled addresses).  Now a fragment read is ONE add of a
// wave-uniform offset to a per-lane base prepared per (tile, column offset), and a loader item is a table entry (LDS offset,
// global offset, row, channel - two registers) + 8 v_perm_b32 + 8 writes with immediate offsets.
template <int NTN, int VEC, int KC, bool EPI = false>   // EPI: the inference epilogue (its own instantiations: the training kernels keep their code)
__global__ __launch_bounds__(kWsThreads) void conv3x3_ws_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w2,
                                                               uint16_t *__restrict__ y, int Cin, int Cout, int NP, int KP, int H,
                                                               int W, int R, int strips, int nblk, int units, int accum,
                                                               const EpiAffine epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int KS = 3, PAD = 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WL = ws_pitch(W), rows_l = R + 2 * PAD, npxl = rows_l * WL;
    const int buf_bytes = KC * npxl * 64;
    // XCD x (workgroup L runs on XCD L % 8) owns a contiguous range of units - neighbouring strips share halo rows through
    // that XCD's L2 - and its workgroups take them round-robin, so strips in flight at the same time are neighbours.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int upx = (units + 7) / 8;
    const int u_end = min(units, (xcd + 1) * upx), first = xcd * upx + slot;
    const int n_mine = first < u_end ? (u_end - first + wpx - 1) / wpx : 0;
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(lds);
    for (int i = tid; i < buf_bytes / 2; i += kWsThreads) lds32[i] = 0u;      // both buffers; the halo columns stay zero
    __syncthreads();
    auto decode = [&](int k, int &b, int &r0, int &nb) {
        const int u = first + k * wpx;
        nb = u % nblk;
        const int bs = u / nblk;
        b = bs / strips;
        r0 = (bs - b * strips) * R;
    };

    if (wave >= 4) {
        // ---------------- loader waves: rows of x -> registers -> LDS records [pixel][32 channels] (slot g ^ 2 if pixel & 4) ----
        const int lt = tid - 256;
        const int nvec_row = W / VEC, nvec = rows_l * nvec_row;
        const int items = 16 * nvec * KC;                        // <= kItems * 256 (launch_conv)
        constexpr int kItems = VEC == 8 ? 8 : 12;
        // item table (the same for every stage and unit): meta = LDS dword offset | channel in the stage << 16 | row << 24
        int meta[kItems], goff[kItems];
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const int it = j * 256 + lt;
            meta[j] = -1;
            goff[j] = 0;
            if (it >= items) continue;
            const int pair = it & 15, vs = it >> 4;
            const int slab = KC == 1 ? 0 : vs / nvec, v = KC == 1 ? vs : vs - slab * nvec;
            const int lr = v / nvec_row, xv = (v - lr * nvec_row) * VEC;
            const int P = lr * WL + xv + PAD;
            // VEC = 8: P = 1 (mod 8), the slot flips of pixels P .. P + 7 are compile-time; VEC = 4: P = 1 or 5 (mod 8), and the
            // pattern of P = 5 is that of P = 1 with every flip inverted - folded into the pair index of the base
            const int pbase = (VEC == 4 && (P & 4)) ? (pair ^ 8) : pair;
            meta[j] = (slab * npxl * 16 + P * 16 + pbase) | ((32 * slab + 2 * pair) << 16) | (lr << 24);
            goff[j] = ((32 * slab + 2 * pair) * H + (lr - PAD)) * W + xv;
        }
        int g = 0;
        for (int k = 0; k < n_mine; ++k) {
            int b, r0, nb;
            decode(k, b, r0, nb);
            const uint16_t *xb = x + (int64_t)b * Cin * H * W + (int64_t)r0 * W;
            for (int c0 = 0; c0 < KP; c0 += 32 * KC, ++g) {
                uint32_t *buf = lds32 + (g & 1) * (buf_bytes / 4);
                const uint16_t *xs = xb + (int64_t)c0 * H * W;
                typename PixVec<VEC>::type v0[kItems], v1[kItems];
#pragma unroll
                for (int j = 0; j < kItems; ++j) {                 // every load of the stage is in flight before the first unpack
                    v0[j] = typename PixVec<VEC>::type{};
                    v1[j] = typename PixVec<VEC>::type{};
                    if (meta[j] < 0 || (kAbl3 & 4)) continue;
                    const int gy = r0 - PAD + (meta[j] >> 24), ca = c0 + ((meta[j] >> 16) & 0xff);
                    if (gy < 0 || gy >= H) continue;             // rows outside the plane are written as zeros (the buffer is reused)
                    if (ca < Cin) v0[j] = *reinterpret_cast<const typename PixVec<VEC>::type *>(xs + goff[j]);
                    if (ca + 1 < Cin) v1[j] = *reinterpret_cast<const typename PixVec<VEC>::type *>(xs + goff[j] + H * W);
                }
#pragma unroll
                for (int j = 0; j < kItems; ++j) {
                    if (meta[j] < 0 || (kAbl3 & 16)) continue;
                    // dword (pair ^ flip) of pixel P + i, flip = 8 where (1 + i) & 4, relative to the base's own flip: two bases, the
                    // per-pixel part is an immediate offset
                    uint32_t *dst0 = buf + (meta[j] & 0xffff), *dst1 = buf + ((meta[j] & 0xffff) ^ 8);
                    const uint32_t *a0 = reinterpret_cast<const uint32_t *>(&v0[j]), *a1 = reinterpret_cast<const uint32_t *>(&v1[j]);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const uint32_t pr = __builtin_amdgcn_perm(a1[i >> 1], a0[i >> 1], (i & 1) ? 0x07060302u : 0x05040100u);
                        (((1 + i) & 4) ? dst1 : dst0)[i * 16] = pr;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        return;
    }

    // ---------------- MFMA waves: wave w -> output channels [16 NTN w, 16 NTN (w + 1)) of the unit's block ----------------
    const int g16 = (lane >> 4) * 16, i16 = lane & 15;
    int pre[kMaxPixTiles][KS];                                   // per pixel tile and column offset: record + swizzled slot, bytes
#pragma unroll
    for (int jt = 0; jt < kMaxPixTiles; ++jt) {
        int q = jt * 16 + i16;
        if (q >= R * W) q = 0;
        const int orow = q / W, ocol = q - orow * W;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int A = (orow * WL + ocol + kx) * 64 + g16;    // bit 8 = bit 2 of the pixel index -> flips bit 5 (slot g ^ 2)
            pre[jt][kx] = A ^ ((A >> 3) & 32);
        }
    }
    // weights: ring of three fragment sets, requested two taps ahead across slabs, stages and units
    auto wptr = [&](int nb) { return w2 + (int64_t)(nb * 64 * NTN + wave * 16 * NTN + i16) * KP + 8 * (lane >> 4); };
    auto load_a = [&](const uint16_t *wp, int tap, bf16x8 (&a)[NTN]) {
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            if (kAbl3 & 8) { a[t] = __builtin_bit_cast(bf16x8, make_uint4(tap, t, lane, 1)); continue; }
            a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(wp + ((int64_t)tap * NP + t * 16) * KP));
        }
    };
    bf16x8 aw[3][NTN];
    if (n_mine > 0) {
        int b, r0, nb;
        decode(0, b, r0, nb);
        load_a(wptr(nb), 0, aw[0]);
        load_a(wptr(nb), 1, aw[1]);
    }
    __syncthreads();                                             // stage 0 of the first unit is in buffer 0
    int g = 0;
    for (int k = 0; k < n_mine; ++k) {
        int b, r0, nb;
        decode(k, b, r0, nb);
        const uint16_t *wu = wptr(nb);
        const uint16_t *wnext_unit = wu;                         // weights of the unit after this one (any valid address at the end)
        if (k + 1 < n_mine) {
            int b2, r2, nb2;
            decode(k + 1, b2, r2, nb2);
            wnext_unit = wptr(nb2);
        }
        f32x4v acc[NTN][kMaxPixTiles];
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int jt = 0; jt < kMaxPixTiles; ++jt) acc[t][jt] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < KP; c0 += 32 * KC, ++g) {
            const unsigned char *buf = lds + (g & 1) * buf_bytes;
            // (Requesting the fragments of the next half-tap in the shadow of the current MFMAs - sched_group_barrier: 2 MFMAs, 1
            // add, 1 ds_read, four reads in flight - changed nothing on the 128-channel blocks and cost 10 % on the 64-channel ones.)
#pragma unroll
            for (int slab = 0; slab < KC; ++slab) {
                const int cs = c0 + 32 * slab;
                if (KC > 1 && cs >= KP) break;
                const uint16_t *wnext = cs + 32 < KP ? wu + cs + 32 : wnext_unit;
#pragma unroll
                for (int jt = 0; jt < kMaxPixTiles; ++jt)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) asm volatile("" : "+v"(pre[jt][kx]));   // (keeps the 90 tap addresses from being formed ahead of the loop)
#pragma unroll
                for (int tap = 0; tap < KS * KS; ++tap) {
                    const unsigned char *rowb = buf + slab * npxl * 64 + (tap / KS) * WL * 64;   // wave-uniform
                    bf16x8 (&a)[NTN] = aw[tap % 3];
                    if (tap + 2 < KS * KS) load_a(wu + cs, tap + 2, aw[(tap + 2) % 3]);
                    else load_a(wnext, tap + 2 - KS * KS, aw[(tap + 2) % 3]);
                    bf16x8 bf[kMaxPixTiles];
#pragma unroll
                    for (int jt = 0; jt < kMaxPixTiles; ++jt) {
                        if (kAbl3 & 2) { bf[jt] = __builtin_bit_cast(bf16x8, make_uint4(pre[jt][0], lane, tap, jt)); continue; }
                        bf[jt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(rowb + pre[jt][tap % KS]));
                    }
                    __builtin_amdgcn_sched_barrier(0);           // requests stay above the MFMAs
                    if (kAbl3 & 1) {
#pragma unroll
                        for (int jt = 0; jt < kMaxPixTiles; ++jt) asm volatile("" ::"v"(bf[jt]));
#pragma unroll
                        for (int t = 0; t < NTN; ++t) asm volatile("" ::"v"(a[t]));
                        continue;
                    }
#pragma unroll
                    for (int jt = 0; jt < kMaxPixTiles; ++jt)
#pragma unroll
                        for (int t = 0; t < NTN; ++t)
                            acc[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bf[jt], acc[t][jt], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        // ---- store through this wave's own LDS tile [16 NTN channels][strip pixels]: a strip is whole rows, i.e. TP contiguous
        // elements per output channel, written with VEC-element stores; the loaders are already filling the next unit's first stage
        const int TP = min(R, H - r0) * W;
        const int n_wave = nb * 64 * NTN + wave * 16 * NTN;
        constexpr int OP = 16 * kMaxPixTiles + 8;
        uint16_t *ot = reinterpret_cast<uint16_t *>(lds + 2 * buf_bytes) + wave * (16 * NTN * OP);
        if (EPI) {                                               // eval-mode BatchNorm / bias + activation on the accumulators
            const float ls = epi.lab ? epi.lab[0] : 1.f, lb = epi.lab ? epi.lab[1] : 0.f;
#pragma unroll
            for (int t = 0; t < NTN; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = min(n_wave + t * 16 + 4 * (lane >> 4) + r, Cout - 1);
                    const float sc = epi.scale[n], sh = epi.shift[n];
#pragma unroll
                    for (int jt = 0; jt < kMaxPixTiles; ++jt)
                        ot[(t * 16 + 4 * (lane >> 4) + r) * OP + jt * 16 + i16] = f32_to_bf16(ls * epi_act(acc[t][jt][r] * sc + sh, epi.act) + lb);
                }
        } else {
#pragma unroll
            for (int t = 0; t < NTN; ++t)
#pragma unroll
                for (int jt = 0; jt < kMaxPixTiles; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[(t * 16 + 4 * (lane >> 4) + r) * OP + jt * 16 + i16] = f32_to_bf16(acc[t][jt][r]);
        }
        uint16_t *yb = y + ((int64_t)b * Cout * H + r0) * W;
        const int cpr = TP / VEC;                                // chunks per channel row (W is a multiple of VEC)
        int row = lane / cpr, c = (lane - row * cpr) * VEC;      // 64 lanes walk the [16 NTN][cpr] chunk grid
        const int drow = 64 / cpr, dc = (64 - drow * cpr) * VEC;
        while (row < 16 * NTN) {
            const int n = n_wave + row;
            if (n < Cout) {
                typename PixVec<VEC>::type v = *reinterpret_cast<const typename PixVec<VEC>::type *>(ot + row * OP + c);
                typename PixVec<VEC>::type *dst = reinterpret_cast<typename PixVec<VEC>::type *>(yb + (int64_t)n * H * W + c);
                if (accum) {                                     // y += conv(x): bf16 + bf16 in fp32, one rounding (like a separate add)
                    const typename PixVec<VEC>::type o = *dst;
                    const uint32_t *a2 = reinterpret_cast<const uint32_t *>(&v), *c2 = reinterpret_cast<const uint32_t *>(&o);
                    uint32_t *r2 = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
                    for (int e = 0; e < VEC / 2; ++e)
                        r2[e] = pack_bf16x2(__uint_as_float(a2[e] << 16) + __uint_as_float(c2[e] << 16),
                                            __uint_as_float(a2[e] & 0xffff0000u) + __uint_as_float(c2[e] & 0xffff0000u));
                }
                *dst = v;
            }
            row += drow;
            c += dc;
            if (c >= TP) { c -= TP; ++row; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolution = per image Y[n, p] = sum_c W[n, c] X[c, p] with X in its NATURAL NCHW layout in LDS:
// rows = channels, pixels contiguous, written with plain 16-byte copies; the MFMA B fragment (8 channels of one
// pixel) comes from two ds_read_b64_tr_b16 (gfx950 LDS transpose-read).  Row pitch = 144 elements (288 B = 32 B
// mod 256 B): the 8 rows a 32-lane half touches fall into 8 different 32-byte bank groups.  The next stage's
// global loads are issued before the MFMAs of the current one (register staging, one LDS buffer).
//   block = 256 threads: 128 pixels x (64 * NTN) output channels; wave w -> channels [16 NTN w, 16 NTN (w + 1)).
typedef short tr_v4s __attribute__((ext_vector_type(4)));
constexpr int kTrPix = 128, kTrPitch = 144;

template <int NTN, int KC, int VEC>
__global__ __launch_bounds__(kConvThreads) void conv1x1_tr_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w2,
                                                                 uint16_t *__restrict__ y, int Cin, int Cout, int NP, int KP,
                                                                 int HW, int ptiles, int total_tiles, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint16_t *tile = reinterpret_cast<uint16_t *>(lds);          // [KC * 32][kTrPitch]
    constexpr int ROWS = KC * 32, VPR = kTrPix / VEC;            // vectors per row
    constexpr int NLD = (ROWS * VPR + kConvThreads - 1) / kConvThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order: workgroup L runs on XCD L % 8 (round-robin dispatch), and every XCD has its own L2.  The
    // nblk output-channel blocks that re-read the same pixel tile of x are made CONSECUTIVE workgroups of ONE XCD,
    // so the tile is fetched from HBM once and re-read from that XCD's L2 (x is re-read nblk = Cout / (64 NTN) times).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nb = slot % nblk, tile_id = (slot / nblk) * 8 + xcd;
    if (tile_id >= total_tiles) return;
    const int b = tile_id / ptiles, pt = tile_id - b * ptiles;
    const int p0 = pt * kTrPix;
    const int npix = min(kTrPix, HW - p0);
    const int ntile = (npix + 15) / 16;
    const uint16_t *xb = x + (int64_t)b * Cin * HW + p0;
    const int n_wave = nb * 64 * NTN + wave * 16 * NTN;
    const int g = lane >> 4, i16 = lane & 15;
    f32x4v acc[NTN][kTrPix / 16];
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
        for (int jt = 0; jt < kTrPix / 16; ++jt) acc[t][jt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    typename PixVec<VEC>::type pf[NLD];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int it = tid + j * kConvThreads;
            const int row = it / VPR, v = it - row * VPR;
            typename PixVec<VEC>::type val{};
            if (it < ROWS * VPR && c0 + row < Cin && v * VEC < npix)
                val = *reinterpret_cast<const typename PixVec<VEC>::type *>(xb + (int64_t)(c0 + row) * HW + v * VEC);
            pf[j] = val;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int it = tid + j * kConvThreads;
            const int row = it / VPR, v = it - row * VPR;
            if (it < ROWS * VPR) *reinterpret_cast<typename PixVec<VEC>::type *>(tile + row * kTrPitch + v * VEC) = pf[j];
        }
    };
    auto load_a = [&](int cs, bf16x8 (&a)[NTN]) {
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            const int n = n_wave + t * 16 + i16;
            uint4 av = make_uint4(0, 0, 0, 0);
            if (n < NP) av = *reinterpret_cast<const uint4 *>(w2 + (int64_t)n * KP + cs + 8 * g);
            a[t] = __builtin_bit_cast(bf16x8, av);
        }
    };
    // per-lane transpose-read base: row 4 g + i/4 of a slab, pixels 4 (i % 4) .. + 3 of a 16-pixel tile
    const int tr_off = (4 * g + (i16 >> 2)) * kTrPitch + 4 * (i16 & 3);

    fetch(0);
    for (int c0 = 0; c0 < KP; c0 += 32 * KC) {
        stash();
        __syncthreads();
        if (c0 + 32 * KC < KP) fetch(c0 + 32 * KC);               // in flight during the MFMAs below
        bf16x8 a[NTN];
        load_a(c0, a);
#pragma unroll 1                                                 // one slab's 16 transpose-reads live at a time
        for (int slab = 0; slab < KC; ++slab) {
            const int cs = c0 + 32 * slab;
            if (KC > 1 && cs >= KP) break;
            bf16x8 an[NTN];
            load_a(min(cs + 32, KP - 32), an);                   // next slab's weights: L2 latency behind this slab's MFMAs
            const uint16_t *sl = tile + slab * 32 * kTrPitch + tr_off;
#pragma unroll
            for (int jt = 0; jt < kTrPix / 16; ++jt) {
                if (jt < ntile) {
                    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (tr_v4s __attribute__((address_space(3))) *)(sl + jt * 16));
                    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (tr_v4s __attribute__((address_space(3))) *)(sl + jt * 16 + 16 * kTrPitch));
                    typedef short tr_v8s __attribute__((ext_vector_type(8)));
                    const tr_v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
                    for (int t = 0; t < NTN; ++t)
                        acc[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bf, acc[t][jt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NTN; ++t) a[t] = an[t];
        }
        __syncthreads();
    }
    uint16_t *yb = y + (int64_t)b * Cout * HW + p0;
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
#pragma unroll
        for (int jt = 0; jt < kTrPix / 16; ++jt) {
            const int q = jt * 16 + i16;
            if (jt < ntile && q < npix) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n_wave + t * 16 + 4 * g + r;
                    if (n < Cout) yb[(int64_t)n * HW + q] = f32_to_bf16(acc[t][jt][r]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolution, second generation: the same LDS transpose-read GEMM fed by asynchronous global -> LDS copies
// (global_load_lds_dwordx4) into a ring of 3 stages, two of them in flight while the third is consumed - the first
// generation (conv1x1_tr_kernel) was bound by the load -> LDS -> barrier latency of ONE register stage (MFMA 18 % busy,
// HBM traffic already minimal: profiles/r01_conv1x1_tr_pmc.txt).
//   stage  = 64 input channels: X [64 ch][128 px] (16 KiB) + W [64 NTN out ch][64 k] (8 NTN KiB), both copied by LDS-DMA.
//            The DMA writes LDS lane-linearly (wave-uniform base + 16 B x lane), so the images are unpadded and the bank
//            conflicts are removed on the SOURCE side: X row r keeps its 16-pixel segment jt at physical segment jt ^ (r & 7)
//            (the 8 rows a 32-lane half transpose-reads then cover all 64 banks), W row n keeps its 16-byte k chunk kc at
//            chunk kc ^ (n & 7) (2-way on the 16-row ds_read_b128).
//   block  = 512 threads = 8 waves (4 over the output channels x 2 over the pixels), tile 128 px x 64 NTN channels,
//            one workgroup per CU (72 / 96 KiB of LDS), two waves per SIMD.
//   sync   = per stage ONE raw s_barrier behind a COUNTED s_waitcnt vmcnt(4): this wave's 4 copies of the stage being
//            consumed have landed, the 4 of the next stage stay in flight across the barrier; the barrier also says every
//            wave is done with the stage consumed before, whose slot the copies issued right after it overwrite.
//   output = accumulators -> bf16 -> per-wave LDS transpose -> 16-byte stores of whole 128-byte pixel rows.
constexpr int kG2Threads = 512, kG2Rows = 64;

// A [B, C, H, W] activation given as up to 8 tensors concatenated along C (each [B, C_k, H, W], contiguous): the 1x1
// kernels read / write the parts in place instead of a torch.cat / split copy (HG_Block aggregation, RepNCSPELAN4.cv4,
// FPN / PAN fusion inputs: src/d_fine/arch/hgnetv2.py:265-274, hybrid_encoder.py:196-206,460-486).
struct ChanSegs {
    const uint16_t *p[8];
    int start[9];          // channel offsets, start[n] = C
    int bs[8];             // batch stride of part k in channels (= its channel count, or more for a channel slice of a wider tensor)
    int n;
};

// address of channel c of image b (per-lane search: epilogues)
__device__ __forceinline__ const uint16_t *seg_addr(const ChanSegs &sg, int b, int c, int HW) {
    const uint16_t *sp = sg.p[0];
    int s0 = 0, sc = sg.bs[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < sg.n && c >= sg.start[k]) { sp = sg.p[k]; s0 = sg.start[k]; sc = sg.bs[k]; }
    return sp + ((int64_t)b * sc + (c - s0)) * HW;
}

// Part table held in registers: the kernel-argument copy would be re-read with scalar loads inside the stage loop, and those
// share lgkmcnt with the LDS reads.
struct SegRegs {
    const uint16_t *p[8];
    int start[8], bs[8], n;
    __device__ __forceinline__ void load(const ChanSegs &sg) {
        n = sg.n;
#pragma unroll
        for (int k = 0; k < 8; ++k) { p[k] = sg.p[k < sg.n ? k : 0]; start[k] = k < sg.n ? sg.start[k] : 0x7fffffff; bs[k] = sg.bs[k < sg.n ? k : 0]; }
    }
    // first channel `cu` (wave-uniform) of a run of rows inside one part -> address of row `row_in_run`
    __device__ __forceinline__ const uint16_t *addr(int b, int cu, int row_in_run, int HW) const {
        const uint16_t *sp = p[0];
        int s0 = 0, sc = bs[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (cu >= start[k]) { sp = p[k]; s0 = start[k]; sc = bs[k]; }
        return sp + ((int64_t)b * sc + (cu - s0) + row_in_run) * HW;
    }
};

// the same for a WAVE-UNIFORM first channel `cu` of a run of rows that lies inside one part (part sizes are multiples of the
// run length): the search runs on the scalar unit, the lanes only add their row offset - this sits in the LDS-DMA issue path
// of every stage, where a per-lane search over 64-bit pointers cost 35 % of the kernel.
__device__ __forceinline__ const uint16_t *seg_addr_uniform(const ChanSegs &sg, int b, int cu, int row_in_run, int HW) {
    const uint16_t *sp = sg.p[0];
    int s0 = 0, sc = sg.bs[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < sg.n && cu >= sg.start[k]) { sp = sg.p[k]; s0 = sg.start[k]; sc = sg.bs[k]; }
    return sp + ((int64_t)b * sc + (cu - s0)) * HW + (int64_t)row_in_run * HW;
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_addr + 16 * lane].  Inline asm on purpose:
// hipcc tracks the builtin form as an LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read, which drains
// the ring every stage; the asm form is invisible to its counters, completion is waited for by hand (counted vmcnt +
// barrier, see the kernel).  M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(const uint16_t *gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

// tools/probe/conv1x1_ablate.hip builds timing-only variants of the kernel with parts of the stage loop removed
// (1: no MFMAs, 2: no LDS fragment reads, 4: no copies after the first two stages and no waiting for them).
#ifndef DFINE_CONV1X1_ABLATE
#define DFINE_CONV1X1_ABLATE 0
#endif
constexpr int kAbl = DFINE_CONV1X1_ABLATE;

template <int NTN, int kG2Ring, int PXW, bool SEG, bool EPI = false>   // EPI: the inference epilogue (own instantiations); kG2Ring LDS stages (3: two in flight; 2 for <= 128 input channels); PXW 16-pixel
                                                     // tiles per wave (4 / 8); SEG: input / output given as several parts
__global__ __launch_bounds__(kG2Threads) void conv1x1_glds_kernel(const ChanSegs xs_, const uint16_t *__restrict__ w2,
                                                                  const ChanSegs ys_, int Cin, int Cout, int NP, int KP,
                                                                  int HW, int ptiles, int total_tiles, int nblk, int accum,
                                                                  int64_t w_bstride /* elements between the images' weight sets: 0 = shared */,
                                                                  int ximg /* tiles over the pixels of ALL images (B * HW), see below */,
                                                                  const EpiAffine epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TP = 32 * PXW;                                             // pixels per workgroup (128 or 256)
    constexpr int XPITCH = TP * 2;                                           // bytes per channel row
    constexpr int XB = kG2Rows * XPITCH, WB = 64 * NTN * 128, SB = XB + WB;  // bytes per stage
    constexpr int LPR = TP / 8, RPP = 64 / LPR, XPW = (kG2Rows / RPP) / 8;   // lanes per row, rows per piece, X pieces per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nb = slot % nblk, tile_id = (slot / nblk) * 8 + xcd;
    if (tile_id >= total_tiles) return;
    // Pixel tiles: per image (ptiles of them, the last one partly empty: 400-pixel planes fill 78 % of four 128-pixel tiles),
    // or - ximg, whole tensors only - over the B * HW pixels of the batch: an 8-pixel chunk (one LDS-DMA piece, one 16-byte
    // store) never straddles two images (HW % 8 == 0), so a lane just carries its own image / pixel base.  `ptiles` then
    // holds the number of images.
    const int b = ximg ? 0 : tile_id / ptiles, pt = ximg ? tile_id : tile_id - b * ptiles;
    const int p0 = ximg ? 0 : pt * TP;
    const int64_t g0 = (int64_t)pt * TP, gtot = (int64_t)ptiles * HW;             // ximg: first pixel of the tile in the batch
    const int npix = ximg ? (int)min((int64_t)TP, gtot - g0) : min(TP, HW - p0);
    const int n0 = nb * 64 * NTN;
    w2 += (int64_t)b * w_bstride;                       // per-image weights (the mask-logit einsum bqc,bchw->bqhw)
    const int wn = wave >> 1, wp = wave & 1;
    const int g = lane >> 4, i16 = lane & 15;
    const int nstage = (KP + kG2Rows - 1) / kG2Rows;

    // ---- per-lane source coordinates of this wave's LDS-DMA pieces (constant over the stages) ----
    // X: 64 / RPP pieces of RPP rows, XPW per wave.  W: 8 NTN pieces of 8 rows; wave -> NTN pieces.
    int x_row[XPW], x_px[XPW], w_row[NTN], w_k[NTN];
    const uint16_t *x_src[XPW];                        // whole-tensor case: this lane's channel-0 address of piece j
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int row = (wave * XPW + j) * RPP + lane / LPR, pc = lane % LPR;
        const int px = (((pc >> 1) ^ (row & 7)) << 4) + ((pc & 1) << 3);
        x_row[j] = row;
        x_px[j] = px < npix ? px : 0;                  // columns past the plane: any valid data, never stored
        if (ximg) {
            const int64_t gp = g0 + x_px[j];
            const int bi = (int)(gp / HW);
            x_src[j] = xs_.p[0] + ((int64_t)bi * xs_.bs[0]) * HW + (gp - (int64_t)bi * HW);
        } else {
            x_src[j] = xs_.p[0] + (int64_t)b * xs_.bs[0] * HW + p0 + x_px[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
        const int row = (wave * NTN + j) * 8 + (lane >> 3), pc = lane & 7;
        w_row[j] = min(n0 + row, NP - 1);               // rows past the layer: any valid row, never stored
        w_k[j] = ((pc ^ (row & 7)) << 3);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    // whole-tensor case: plain pointer arithmetic - the part table lives in the kernel-argument segment, and scalar loads from
    // it inside the stage loop share lgkmcnt with the LDS reads (every wait for one drains the other)
    // several parts: a per-workgroup table in LDS, one 8-byte base address per group of 8 input channels (part sizes are
    // multiples of 8), built once; the issue path then costs one ds_read_b64 per piece.  (Searching the part list there - per
    // lane or on the scalar unit, from kernel arguments or registers - cost 2x of the whole kernel.)
    // (the tables sit behind the ring AND behind the epilogue's staging tiles, which are larger than a 2-slot ring for NTN = 4)
    constexpr int kTabOff = kG2Ring * SB > 8 * 16 * NTN * (16 * PXW + 8) * 2 ? kG2Ring * SB : 8 * 16 * NTN * (16 * PXW + 8) * 2;
    const uint16_t **xtab = reinterpret_cast<const uint16_t **>(lds + kTabOff);
    // ... and one per output row of this workgroup for the store loop (a per-lane search of the part list there was ~100
    // vector instructions per 16-byte store, ~800 per wave and tile after the last MFMA)
    uint16_t **ytab = reinterpret_cast<uint16_t **>(lds + kTabOff + 4096);
    if (SEG) {
        for (int gch = tid; gch * 8 < Cin; gch += kG2Threads) xtab[gch] = seg_addr(xs_, b, gch * 8, HW) + p0;
        if (tid < 64 * NTN) ytab[tid] = const_cast<uint16_t *>(seg_addr(ys_, b, min(n0 + tid, Cout - 1), HW));
        __syncthreads();
    }
    auto issue = [&](int s) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (s % kG2Ring) * SB);
        const int c0 = s * kG2Rows;
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            if (!SEG) {
                const int ch = min(c0 + x_row[j], Cin - 1);
                glds16(x_src[j] + (int64_t)ch * HW, __builtin_amdgcn_readfirstlane(base + (wave * XPW + j) * 1024));
                continue;
            }
            const int ch = min(c0 + x_row[j], Cin - 1);  // channels past Cin: any valid row - they meet zero weights or a skipped slab
            glds16(xtab[ch >> 3] + (int64_t)(ch & 7) * HW + x_px[j], __builtin_amdgcn_readfirstlane(base + (wave * XPW + j) * 1024));
        }
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const int k = min(c0 + w_k[j], KP - 8);      // k chunks past KP belong to a slab that is skipped
            glds16(w2 + (int64_t)w_row[j] * KP + k, __builtin_amdgcn_readfirstlane(base + XB + (wave * NTN + j) * 1024));
        }
    };

    f32x4v acc[NTN][PXW];
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
        for (int j = 0; j < PXW; ++j) acc[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // NTN = 4 (256 output channels x 256 pixels per workgroup: 128 FLOP per loaded byte instead of 85): the stage is 64 KiB, so
    // the ring has TWO slots and is streamed - stage s + 1 is issued behind the barrier of stage s into the slot stage s - 1 has
    // just released - and the fragments are read one 32-channel slab at a time (128 accumulator + 48 fragment registers).
    // Measured (tools/conv1x1_bench.py, forward): 512 -> 512 @80x80 202 -> 158 us (679 TFLOP/s), @40x40 53.5 -> 42.2,
    // 768 -> 256 @80x80 144 -> 124, 384 -> 768 @40x40 60 -> 52.6; 26 layer shapes 1364 -> 1259 us.  One stage (64 KiB per CU) in
    // flight is what LDS allows here and it does not cover the load latency (a stage takes ~7 600 cycles for 2 048 cycles of
    // MFMA work); the early / late issue split of csrc/wgrad3.hip needs a third slot (late waves would wait for loads they
    // have just issued: measured 172 us).  Also measured and dropped: 32-channel HALF stages of 32 KiB in a ring of four slots,
    // three in flight, with the early / late split (W rows of 64 B, chunk kc of row r at kc ^ ((r >> 2) & 3)): 167.6 us, 26 shapes
    // 1311 us - more bytes in flight do not help, so the stage time is not exposed load latency: MFMA (2 048 cycles per SIMD and
    // stage), LDS-DMA issue (~1 900) and fragment reads (~770) run one after the other inside a wave, and the barrier per
    // stage keeps the two waves of a SIMD in phase.
    const bool kStream2 = kG2Ring == 2 && (NTN == 4 || nstage > 2);     // (2-slot ring with more than two stages: streamed)

    issue(0);
    if (!kStream2 && nstage > 1) issue(1);
    for (int s = 0; s < nstage; ++s) {
        if (kStream2) {
            if (!(kAbl & 4) || s < 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char *xs2 = lds + (s & 1) * SB, *ws2 = xs2 + XB;
            // (measured and dropped: the copies of stage s + 1 spread one piece behind every other MFMA group of the stage,
            // order pinned by sched_barrier - 26 shapes 1161 -> 1218 us: the copies queueing at the CU's address path in front of
            // the first MFMA are not what the stage waits for)
            const bool more = s + 1 < nstage && !((kAbl & 4) && s >= 1);
            if (more) issue(s + 1);
            const int nslab = s * kG2Rows + 32 < KP ? 2 : 1;
#pragma unroll
            for (int slab = 0; slab < 2; ++slab) {
                if (slab >= nslab) break;
                bf16x8 a1[NTN], b1[PXW];
#pragma unroll
                for (int t = 0; t < NTN; ++t) {
                    const int row = wn * 16 * NTN + t * 16 + i16;
                    if (kAbl & 2) { a1[t] = __builtin_bit_cast(bf16x8, make_uint4(row, lane, s, t)); continue; }
                    a1[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(ws2 + row * 128 + (((slab * 4 + g) ^ (row & 7)) << 4)));
                }
                const int r = slab * 32 + 4 * g + (i16 >> 2);
                const unsigned char *xr = xs2 + r * XPITCH + ((i16 & 3) << 3);
#pragma unroll
                for (int j = 0; j < PXW; ++j) {
                    const int seg = ((wp * PXW + j) ^ (r & 7)) << 5;
                    if (kAbl & 2) { b1[j] = __builtin_bit_cast(bf16x8, make_uint4(seg, lane, s, j)); continue; }
                    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_v4s __attribute__((address_space(3))) *)(xr + seg));
                    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_v4s __attribute__((address_space(3))) *)(xr + seg + 16 * XPITCH));
                    typedef short tr_v8s __attribute__((ext_vector_type(8)));
                    b1[j] = __builtin_bit_cast(bf16x8, (tr_v8s)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                if (kAbl & 1) {                                   // fragments kept alive, no matrix work
#pragma unroll
                    for (int t = 0; t < NTN; ++t) asm volatile("" ::"v"(a1[t]));
#pragma unroll
                    for (int j = 0; j < PXW; ++j) asm volatile("" ::"v"(b1[j]));
                    continue;
                }
#pragma unroll
                for (int j = 0; j < PXW; ++j) {
#pragma unroll
                    for (int t = 0; t < NTN; ++t)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[t], b1[j], acc[t][j], 0, 0, 0);
                }
            }
            continue;
        }
        if (!(kAbl & 4) || s < 2) {
            if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XPW + NTN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char *xs = lds + (s % kG2Ring) * SB, *ws = xs + XB;
        // Every fragment of the stage's two 32-channel slabs is requested BEFORE the first MFMA (the loop this replaces read
        // one B fragment, waited for it, issued its NTN MFMAs, read the next: ~150 cycles of exposed LDS latency per fragment,
        // 16 x per stage - the kernel ran at a fifth of its MFMA time, profiles/r03_conv1x1_schedule.txt); the copies of stage
        // s + 2 are issued behind the reads, so their issue slots overlap the LDS latency instead of preceding it.
        const bool two = s * kG2Rows + 32 < KP;               // uniform: the second slab exists (KP is a multiple of 32)
        bf16x8 a[2][NTN], bq[2][PXW];
#pragma unroll
        for (int slab = 0; slab < 2; ++slab) {
#pragma unroll
            for (int t = 0; t < NTN; ++t) {
                const int row = wn * 16 * NTN + t * 16 + i16;
                if (kAbl & 2) { a[slab][t] = __builtin_bit_cast(bf16x8, make_uint4(row, lane, s, t)); continue; }
                a[slab][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(ws + row * 128 + (((slab * 4 + g) ^ (row & 7)) << 4)));
            }
            const int r = slab * 32 + 4 * g + (i16 >> 2);
            const unsigned char *xr = xs + r * XPITCH + ((i16 & 3) << 3);
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                const int seg = ((wp * PXW + j) ^ (r & 7)) << 5;
                if (kAbl & 2) { bq[slab][j] = __builtin_bit_cast(bf16x8, make_uint4(seg, lane, s, j)); continue; }
                const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_v4s __attribute__((address_space(3))) *)(xr + seg));
                const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_v4s __attribute__((address_space(3))) *)(xr + seg + 16 * XPITCH));
                typedef short tr_v8s __attribute__((ext_vector_type(8)));
                bq[slab][j] = __builtin_bit_cast(bf16x8, (tr_v8s)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
        if (kG2Ring >= 3 && s + 2 < nstage && !(kAbl & 4)) issue(s + 2);      // ring of 2: launched for <= 2 stages only, both issued above
        if (kAbl & 1) {                                        // fragments kept alive, no matrix work
#pragma unroll
            for (int slab = 0; slab < 2; ++slab) {
#pragma unroll
                for (int t = 0; t < NTN; ++t) asm volatile("" ::"v"(a[slab][t]));
#pragma unroll
                for (int j = 0; j < PXW; ++j) asm volatile("" ::"v"(bq[slab][j]));
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < PXW; ++j)
#pragma unroll
            for (int t = 0; t < NTN; ++t)
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][t], bq[0][j], acc[t][j], 0, 0, 0);
        if (two) {
#pragma unroll
            for (int j = 0; j < PXW; ++j)
#pragma unroll
                for (int t = 0; t < NTN; ++t)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][t], bq[1][j], acc[t][j], 0, 0, 0);
        }
    }
    // ---- epilogue: [n][px] bf16 tile of this wave through LDS, then whole 128-byte pixel rows with 16-byte stores ----
    __builtin_amdgcn_s_barrier();                        // every wave is done reading the ring
    constexpr int OP = 16 * PXW + 8;                     // output tile pitch (elements)
    uint16_t *ot = reinterpret_cast<uint16_t *>(lds) + wave * (16 * NTN * OP);          // [16 NTN rows][16 PXW px]
    if (EPI) {                                           // eval-mode BatchNorm / bias + activation on the accumulators, row by row
        const float ls = epi.lab ? epi.lab[0] : 1.f, lb = epi.lab ? epi.lab[1] : 0.f;     // (a separate copy of the store loop: the
#pragma unroll                                                                             //  training launches keep their registers)
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = min(n0 + wn * 16 * NTN + t * 16 + 4 * g + r, Cout - 1);
                const float sc = epi.scale[n], sh = epi.shift[n];
#pragma unroll
                for (int j = 0; j < PXW; ++j)
                    ot[(t * 16 + 4 * g + r) * OP + j * 16 + i16] = f32_to_bf16(ls * epi_act(acc[t][j][r] * sc + sh, epi.act) + lb);
            }
    } else {
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int j = 0; j < PXW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[(t * 16 + 4 * g + r) * OP + j * 16 + i16] = f32_to_bf16(acc[t][j][r]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int LPO = 2 * PXW, RPI = 64 / LPO;         // lanes per output row, rows per iteration
    const int c8l = wp * 16 * PXW + (lane % LPO) * 8;    // this lane's 8-pixel chunk of the tile (the same for every row)
    int64_t ylane = (int64_t)b * ys_.bs[0] * HW + p0 + c8l;      // whole-tensor case: element offset of channel 0 of that chunk
    if (ximg) {
        const int64_t gp = g0 + (c8l < npix ? c8l : 0);
        const int bi = (int)(gp / HW);
        ylane = ((int64_t)bi * ys_.bs[0]) * HW + (gp - (int64_t)bi * HW);
    }
#pragma unroll
    for (int it = 0; it < 16 * NTN / RPI; ++it) {
        const int row = it * RPI + lane / LPO, c8 = (lane % LPO) * 8;
        const int n = n0 + wn * 16 * NTN + row;
        if (n < Cout && wp * 16 * PXW + c8 < npix) {
            uint16_t *yp = SEG ? ytab[wn * 16 * NTN + row] + p0 + wp * 16 * PXW + c8
                               : const_cast<uint16_t *>(ys_.p[0]) + ylane + (int64_t)n * HW;
            bool acc_row = accum != 0;
            if (SEG) {          // several output parts: bit 0 of a part's base address says "add onto this part" (launch_conv1x1)
                const uintptr_t u = reinterpret_cast<uintptr_t>(yp);
                acc_row = acc_row || (u & 1);
                yp = reinterpret_cast<uint16_t *>(u & ~(uintptr_t)1);
            }
            uint4 v = *reinterpret_cast<const uint4 *>(ot + row * OP + c8);
            if (acc_row) {                                // y += conv(x): bf16 + bf16 in fp32, one rounding (like a separate add)
                const uint4 o = *reinterpret_cast<const uint4 *>(yp);
                const uint32_t a[4] = {v.x, v.y, v.z, v.w}, c[4] = {o.x, o.y, o.z, o.w};
                uint32_t r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    r[k] = pack_bf16x2(__uint_as_float(a[k] << 16) + __uint_as_float(c[k] << 16),
                                       __uint_as_float(a[k] & 0xffff0000u) + __uint_as_float(c[k] & 0xffff0000u));
                v = make_uint4(r[0], r[1], r[2], r[3]);
            }
            *reinterpret_cast<uint4 *>(yp) = v;
        }
    }
}

// shapes the LDS-DMA kernel takes (the others run on conv1x1_tr_kernel: whole tensors, no accumulation)
static bool conv1x1_glds_ok(int Cin, int KP, int HW) {
    static const int v2_env = [] { const char *e = getenv("DFINE_CONV1X1_GLDS"); return e ? atoi(e) : 1; }();
    return v2_env && HW % 8 == 0 && KP >= 8 && Cin >= 4 && Cin % 4 == 0;
}

static ChanSegs one_seg(const void *p, int C) {
    ChanSegs sg{};
    sg.p[0] = (const uint16_t *)p; sg.start[0] = 0; sg.start[1] = C; sg.bs[0] = C; sg.n = 1;
    return sg;
}

// tile choice of the LDS-DMA 1x1 kernel for one layer
struct C1Cfg { bool n256, px256, wide2, ring2, ximg; int tp, ptiles2, total2, nblk2; };
static C1Cfg conv1x1_cfg(int B, int NP, int KP, int HW, bool seg, bool per_image_weights) {
    // The kernel is bound by the ~10 B/clk/CU load path, so the tile is as large as the layer can fill the chip with:
    // 256 pixels x 128 channels (87 FLOP per loaded byte) for deep layers on big maps, 128 x 128 (64) / 128 x 64 otherwise.
    // Three tile regimes (tools/conv1x1_bench.py, 26 layer shapes of D-FINE-m, forward: 1364 us with the round-4 choice ->
    // 1193-1259 with either new one alone; per layer the better of the two):
    //  * <= 2 stages (K <= 128): 2-slot ring, both stages issued up front, 2 workgroups per CU (as before);
    //  * deep K (>= 512) and output channels a multiple of 256 with enough such tiles: 256 channels x 256 pixels (NTN = 4),
    //    2-slot ring of 64 KiB stages, streamed - 128 FLOP per loaded byte;
    //  * everything else: 128 (64) channels x 128 pixels with the 2-slot ring STREAMED (64 KiB of LDS: two workgroups = 16
    //    waves per CU, which cover each other's load -> MFMA -> store phases; the round-4 choice for these layers was one
    //    256-pixel workgroup per CU on a 3-slot ring: 1280 -> 384 @40x40 86 -> 70 us, 1792 -> 768 @20x20 68 -> 54, 192 -> 384
    //    @80x80 71 -> 57, 896 -> 384 @40x40 59 -> 49).
    C1Cfg c;
    const bool classic2 = KP <= 2 * kG2Rows;
    const bool n128 = NP % 128 == 0;
    // whole tensors with shared weights: pixel tiles over the B * HW pixels of the batch whenever per-image tiles would
    // leave the last one partly empty (20 x 20 planes: 400 pixels = 3.1 tiles of 128; 40 x 40: 6.25 tiles of 256)
    const bool xok = !seg && !per_image_weights;
    const int64_t gpix = (int64_t)B * HW;
    const int64_t t256 = xok ? (gpix + 255) / 256 : (int64_t)B * ((HW + 255) / 256);
    const bool px256c = n128 && !classic2 && (xok || HW % 256 == 0 || HW >= 1536) && t256 * (NP / 128) >= 256;
    const int64_t blk256 = t256 * (NP / 256);
    c.n256 = px256c && NP % 256 == 0 && KP >= 512 && blk256 >= 200 && !(blk256 > 256 && blk256 < 384);
    const bool small = !c.n256 && !classic2;
    c.ring2 = classic2 || small;
    c.px256 = c.n256 || (px256c && !small);
    c.tp = c.px256 ? 256 : kTrPix;
    c.ximg = xok && HW % c.tp != 0;
    c.ptiles2 = c.ximg ? B : (HW + c.tp - 1) / c.tp;     // (ximg: the kernel wants the image count here)
    c.total2 = c.ximg ? (int)((gpix + c.tp - 1) / c.tp) : B * c.ptiles2;
    c.wide2 = c.px256 || (n128 && ((int64_t)c.total2 * (NP / 128) >= 256));
    c.nblk2 = c.n256 ? NP / 256 : c.wide2 ? NP / 128 : (NP + 63) / 64;
    return c;
}

static int launch_conv1x1(const uint16_t *x, const uint16_t *w2, uint16_t *y, int B, int Cin, int Cout, int NP, int KP, int HW,
                          hipStream_t st, const ChanSegs *xsegs = nullptr, const ChanSegs *ysegs = nullptr, int accum = 0,
                          int64_t w_bstride = 0, unsigned accum_parts = 0 /* bit k: add onto output part k (several parts) */,
                          const EpiAffine epi = EpiAffine{nullptr, nullptr, nullptr, 0}) {
    const int ptiles = (HW + kTrPix - 1) / kTrPix;
    if (conv1x1_glds_ok(Cin, KP, HW)) {
        const ChanSegs xs_ = xsegs ? *xsegs : one_seg(x, Cin);
        ChanSegs ys_ = ysegs ? *ysegs : one_seg(y, Cout);
        const bool seg = xs_.n > 1 || ys_.n > 1;
        if (accum_parts) {
            if (!seg) accum = 1;                                // one part in, one part out: the plain accumulate-into launch
            else
                for (int k = 0; k < ys_.n; ++k)
                    if (accum_parts >> k & 1) ys_.p[k] = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(ys_.p[k]) | 1);
        }
        const C1Cfg cf = conv1x1_cfg(B, NP, KP, HW, seg, w_bstride != 0);
        const bool n256 = cf.n256, px256 = cf.px256, wide2 = cf.wide2, ring2 = cf.ring2, ximg = cf.ximg;
        const int tp = cf.tp, ptiles2 = cf.ptiles2, total2 = cf.total2, nblk2 = cf.nblk2;
        dim3 grid2(8 * ((total2 + 7) / 8) * nblk2);
        if (seg && Cin > 4096) return DFINE_E_BADARG;
        const size_t lds2 = n256 ? (size_t)8 * 64 * 136 * 2 + (seg ? 6144 : 0)
                                 : (size_t)(ring2 ? 2 : 3) * (kG2Rows * tp * 2 + 64 * (wide2 ? 2 : 1) * 128) + (seg ? 5120 : 0);
        static bool attr2 = false;
        if (!attr2) {
            hipError_t e = hipSuccess, r;
#define DFINE_G2_ATTR1(N, R, P, S, BYTES) \
    if ((r = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_glds_kernel<N, R, P, S, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)) != hipSuccess) e = r; \
    if ((r = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_glds_kernel<N, R, P, S, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES)) != hipSuccess) e = r;
#define DFINE_G2_ATTR(N, R, P) \
    DFINE_G2_ATTR1(N, R, P, false, R * (64 * 64 * P + 8192 * N)) DFINE_G2_ATTR1(N, R, P, true, R * (64 * 64 * P + 8192 * N) + 5120)
            DFINE_G2_ATTR(1, 2, 4) DFINE_G2_ATTR(1, 3, 4) DFINE_G2_ATTR(2, 2, 4) DFINE_G2_ATTR(2, 3, 4) DFINE_G2_ATTR(2, 3, 8)
#undef DFINE_G2_ATTR
            DFINE_G2_ATTR1(4, 2, 8, false, 8 * 64 * 136 * 2) DFINE_G2_ATTR1(4, 2, 8, true, 8 * 64 * 136 * 2 + 6144)
#undef DFINE_G2_ATTR1
            if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
            attr2 = true;
        }
#define DFINE_G2LE(N, R, P, S, E) \
    hipLaunchKernelGGL((conv1x1_glds_kernel<N, R, P, S, E>), grid2, dim3(kG2Threads), lds2, st, xs_, w2, ys_, Cin, Cout, NP, KP, HW, ptiles2, total2, nblk2, accum, w_bstride, ximg ? 1 : 0, epi)
#define DFINE_G2L(N, R, P, S) { if (epi.scale) DFINE_G2LE(N, R, P, S, true); else DFINE_G2LE(N, R, P, S, false); }
#define DFINE_G2(N, R, P) { if (seg) DFINE_G2L(N, R, P, true) else DFINE_G2L(N, R, P, false) }
        if (n256) DFINE_G2(4, 2, 8)
        else if (px256) DFINE_G2(2, 3, 8)
        else if (wide2) { if (ring2) DFINE_G2(2, 2, 4) else DFINE_G2(2, 3, 4) }
        else { if (ring2) DFINE_G2(1, 2, 4) else DFINE_G2(1, 3, 4) }
#undef DFINE_G2
#undef DFINE_G2L
#undef DFINE_G2LE
        return check_launch();
    }
    if (xsegs || ysegs || accum || accum_parts || w_bstride || epi.scale) return DFINE_E_BADARG;  // the first-generation kernel takes whole tensors, shared weights, no accumulation, no epilogue
    const int vec = (HW % 8 == 0) ? 8 : (HW % 4 == 0 ? 4 : 2);
    int kc = KP >= 128 ? 4 : (KP >= 64 ? 2 : 1);
    while (kc > 1 && KP < 32 * kc) kc >>= 1;
    const bool wide = (NP % 128 == 0) && ((int64_t)B * ptiles * (NP / 128) >= 512);
    const int nblk = wide ? NP / 128 : (NP + 63) / 64;
    const int total_tiles = B * ptiles;
    dim3 grid(8 * ((total_tiles + 7) / 8) * nblk);
    const size_t ldsb = (size_t)kc * 32 * kTrPitch * 2;
#define DFINE_TR(NTNN, KCC, VECC)                                                                            \
    hipLaunchKernelGGL((conv1x1_tr_kernel<NTNN, KCC, VECC>), grid, dim3(kConvThreads), ldsb, st, x, w2, y, Cin, Cout, NP, \
                       KP, HW, ptiles, total_tiles, nblk)
#define DFINE_TR_K(NTNN, VECC)                                                                               \
    { if (kc == 4) DFINE_TR(NTNN, 4, VECC); else if (kc == 2) DFINE_TR(NTNN, 2, VECC); else DFINE_TR(NTNN, 1, VECC); }
#define DFINE_TR_V(NTNN)                                                                                     \
    { if (vec == 8) DFINE_TR_K(NTNN, 8) else if (vec == 4) DFINE_TR_K(NTNN, 4) else DFINE_TR_K(NTNN, 2) }
    if (wide) DFINE_TR_V(2) else DFINE_TR_V(1)
#undef DFINE_TR_V
#undef DFINE_TR_K
#undef DFINE_TR
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[n][c][tap] = sum_{b, p} dY[b][n][p] * X[b][c][p + shift(tap)].
// GEMM view: M = output channels, N = input channels, K = pixels - in NCHW BOTH operands are
// contiguous along K, so fragments are plain 16-byte reads: dY straight from global, X from an LDS
// copy of the strip (+halo, zero padded) in its natural [channel][row][col] layout.  The +-1 column
// shifts of a 3x3 kernel are made from the aligned 16-byte chunk plus one neighbouring dword with
// v_alignbit (no unaligned LDS access).  A block owns a 64 x 64 (n, c) tile pair for a range of
// (image, strip) units and writes fp32 partial sums; conv_wgrad_reduce_kernel adds the splits.
// A 3x3 kernel is split by kernel ROW over blockIdx.z (3 x the blocks for the same partial-sum buffer,
// no row halo in LDS, 3 instead of 9 accumulator sets); the next unit's X slab is prefetched into registers
// while the MFMAs of the current one run, and the dY fragment of the next K step is loaded one step ahead.
template <int KS>
__global__ __launch_bounds__(kConvThreads, 2) void conv_wgrad_kernel(      // two workgroups per CU: at most 256 registers
    const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy, float *__restrict__ part, int Cin,
    int Cout, int H, int W, int R, int strips, int total_units, int units_per_split, int nct64, int NP16,
    int CP16, int CS /* LDS elements per channel: = 8 (mod 128) -> the 16 channel lanes of a b128 read hit 16 distinct 16-byte bank slots */,
    int npairs, int nsplits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PAD = KS / 2;
    constexpr int LPAD = KS == 3 ? 8 : 0;
    constexpr int TAPS = KS * KS;
    constexpr int NPF = 5;                                       // 64 ch x <= 160 px / 8 / 256 threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order (workgroup L runs on XCD L % 8, each XCD has its own L2): all (n, c) tile pairs of ONE split -
    // the blocks that read the same pixels of x and dY - are consecutive workgroups of one XCD, so those pixels come
    // from HBM once.  With the pair index fastest instead, an XCD sees one c tile with every n tile and dY is fetched
    // once per XCD: 1.9 GB instead of 0.42 GB on the 512 x 512 @ 80x80 layer (FETCH_SIZE, profiles/r01_conv_pmc.txt).
    // (used when the split count is a multiple of 8 - wgrad_plan rounds it - so that every XCD gets the same work)
    int pair, split;
    if ((nsplits & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pair = slot % npairs; split = (slot / npairs) * 8 + xcd;
    } else {
        pair = blockIdx.x % npairs; split = blockIdx.x / npairs;
    }
    if (split >= nsplits) return;
    const int nt64 = pair / nct64, ct64 = pair - nt64 * nct64;
    const int kr = blockIdx.z;                                   // kernel row handled by this block
    const int PW = W + 2 * LPAD;
    uint16_t *xs = reinterpret_cast<uint16_t *>(lds);            // [64][R][PW]
    {   // zero once: the pad columns are never written again
        uint32_t *z = reinterpret_cast<uint32_t *>(lds);
        for (int i = tid; i < 64 * CS / 2; i += kConvThreads) z[i] = 0u;
    }
    f32x4v acc[4][KS];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < KS; ++t) acc[ct][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int n_lane = nt64 * 64 + wave * 16 + (lane & 15);
    const int c_base = ct64 * 64;
    const int nv = W / 8;
    const int nstage = 64 * R * nv;

    // per-thread slab coordinates are the same for every unit
    int st_c[NPF], st_lr[NPF], st_xv[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int it = tid + j * kConvThreads;
        const int ch = it / (R * nv), rem = it - ch * (R * nv);
        st_c[j] = ch; st_lr[j] = rem / nv; st_xv[j] = (rem - st_lr[j] * nv) * 8;
    }
    constexpr int NKS = KS == 3 ? 5 : 1;                         // K steps of 32 pixels per unit (<= 160 pixels)
    uint4 pf[NPF], apf[NKS];
    auto fetch_unit = [&](int u) {
        const int b = u / strips, strip = u - b * strips;
        const int r0 = strip * R;
        const uint16_t *xb = x + (int64_t)b * Cin * H * W;
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            pf[j] = make_uint4(0, 0, 0, 0);
            const int gy = r0 + st_lr[j] + kr - PAD, c = c_base + st_c[j];
            if (tid + j * kConvThreads < nstage && gy >= 0 && gy < H && c < Cin)
                pf[j] = *reinterpret_cast<const uint4 *>(xb + ((int64_t)c * H + gy) * W + st_xv[j]);
        }
        if (KS == 3) {
            // 3x3: the unit's dY fragments (<= 160 pixels = 5 K steps) are fetched a whole unit ahead like the X slab
            // (the 1x1 kernel measured SLOWER with this front-loaded fetch and keeps its one-step-ahead loads)
            const int tpu = min(R, H - r0) * W;
            const uint16_t *dyu = dy + ((int64_t)b * Cout * H + r0) * W;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int px = ks * 32 + 8 * (lane >> 4);
                apf[ks] = make_uint4(0, 0, 0, 0);
                if (px < tpu && n_lane < Cout) apf[ks] = *reinterpret_cast<const uint4 *>(dyu + (int64_t)n_lane * H * W + px);
            }
        }
    };

    // LDS element offset of this lane's fragment per K step (the same for every unit; pixels past a short last strip meet a
    // zero dY fragment, whatever finite values the slab still holds there)
    int poff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int px = min(ks * 32 + 8 * (lane >> 4), R * W - 8);
        const int row = px / W, col = px - row * W;
        poff[ks] = (lane & 15) * CS + row * PW + LPAD + col;
    }
    const int u0 = split * units_per_split, u1 = min(total_units, u0 + units_per_split);
    if (u0 < u1) fetch_unit(u0);
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
        const int b = u / strips, strip = u - b * strips;
        const int r0 = strip * R;
        const int rows = min(R, H - r0);
#pragma unroll
        for (int j = 0; j < NPF; ++j)
            if (tid + j * kConvThreads < nstage)
                *reinterpret_cast<uint4 *>(xs + (st_c[j] * CS + st_lr[j] * PW + LPAD + st_xv[j])) = pf[j];
        uint4 acur[NKS];
        if (KS == 3) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) acur[ks] = apf[ks];
        }
        __syncthreads();
        if (u + 1 < u1) fetch_unit(u + 1);                         // in flight during the MFMAs below
        const uint16_t *dyb = dy + ((int64_t)b * Cout * H + r0) * W;
        const int tp = rows * W;
        auto load_a = [&](int k0) {
            const int px = k0 + 8 * (lane >> 4);
            uint4 av = make_uint4(0, 0, 0, 0);
            if (px < tp && n_lane < Cout) av = *reinterpret_cast<const uint4 *>(dyb + (int64_t)n_lane * H * W + px);
            return av;
        };
        if (KS == 3) {
            // 3x3: the strip is <= 160 pixels = NKS static K steps, two channel tiles per request group, the next group in
            // flight during the MFMAs of the current one.  The +-1 column shifts are made from the aligned 16-byte fragment and
            // one neighbouring dword each side with v_alignbit (8 per 3 MFMAs).  (Unaligned 16-byte LDS reads work on gfx950 but
            // are served in many passes: the kernel ran 2.6 x SLOWER with them.  The dynamic K-step index of the loop this
            // replaces cost register-select chains and a division per step: 7.9 vector instructions per MFMA.)
            uint4 fc[2][2];
            uint32_t fl[2][2], fr[2][2];
            auto read3 = [&](int hs, uint4 (&c)[2], uint32_t (&l)[2], uint32_t (&r)[2]) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const uint16_t *e = xs + ((2 * (hs & 1) + q) * 16 * CS + poff[hs >> 1]);
                    c[q] = *reinterpret_cast<const uint4 *>(e);
                    l[q] = *reinterpret_cast<const uint32_t *>(e - 2);
                    r[q] = *reinterpret_cast<const uint32_t *>(e + 8);
                }
            };
            read3(0, fc[0], fl[0], fr[0]);
#pragma unroll
            for (int hs = 0; hs < 2 * NKS; ++hs) {
                if ((hs >> 1) * 32 >= tp) break;
                if (hs + 1 < 2 * NKS && ((hs + 1) >> 1) * 32 < tp) read3(hs + 1, fc[(hs + 1) & 1], fl[(hs + 1) & 1], fr[(hs + 1) & 1]);
                const bf16x8 a = __builtin_bit_cast(bf16x8, acur[hs >> 1]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ct = 2 * (hs & 1) + q;
                    const uint4 c1 = fc[hs & 1][q];
                    const uint32_t lw = fl[hs & 1][q], rw = fr[hs & 1][q];
                    uint4 b0, b2;
                    b0.x = (lw >> 16) | (c1.x << 16); b0.y = (c1.x >> 16) | (c1.y << 16);
                    b0.z = (c1.y >> 16) | (c1.z << 16); b0.w = (c1.z >> 16) | (c1.w << 16);
                    b2.x = (c1.x >> 16) | (c1.y << 16); b2.y = (c1.y >> 16) | (c1.z << 16);
                    b2.z = (c1.z >> 16) | (c1.w << 16); b2.w = (c1.w >> 16) | (rw << 16);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b0), acc[ct][0], 0, 0, 0);
                    acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, c1), acc[ct][1], 0, 0, 0);
                    acc[ct][KS - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b2), acc[ct][KS - 1], 0, 0, 0);
                }
            }
            __syncthreads();
            continue;
        }
        uint4 av = load_a(0);
        uint4 fc[2][4];
        auto read_step = [&](int k0, uint4 (&c1)[4]) {
            const int px = k0 + 8 * (lane >> 4);
            const int pxc = px < tp ? px : 0;
            const int row = pxc / W, col = pxc - row * W;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                c1[ct] = *reinterpret_cast<const uint4 *>(xs + ((ct * 16 + (lane & 15)) * CS + row * PW + LPAD + col));
        };
        auto mfma_step = [&](const bf16x8 a, const uint4 (&c1v)[4]) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, c1v[ct]), acc[ct][0], 0, 0, 0);
        };
        read_step(0, fc[0]);
        for (int k0 = 0; k0 < tp; k0 += 64) {                       // two K steps per trip: static register sets
            const uint4 an = load_a(k0 + 32);                       // zero beyond the strip
            const bool second = k0 + 32 < tp;
            if (second) read_step(k0 + 32, fc[1]);
            mfma_step(__builtin_bit_cast(bf16x8, av), fc[0]);
            if (!second) break;
            const uint4 an2 = load_a(k0 + 64);
            if (k0 + 64 < tp) read_step(k0 + 64, fc[0]);
            mfma_step(__builtin_bit_cast(bf16x8, an), fc[1]);
            av = an2;
        }
        __syncthreads();
    }
    // ---- partial sums: part[split][n][c][tap], this block's taps = kr * KS .. + KS - 1 ------------
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int c = c_base + ct * 16 + (lane & 15);
        if (c >= CP16) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nt64 * 64 + wave * 16 + 4 * (lane >> 4) + r;
            if (n >= NP16) continue;
            float *dst = part + (((int64_t)split * NP16 + n) * CP16 + c) * TAPS + kr * KS;
#pragma unroll
            for (int t = 0; t < KS; ++t) dst[t] = acc[ct][t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 weight gradient, second generation.  dW[n][c] = sum_{b, p} dY[b][n][p] X[b][c][p] is a GEMM with a 128-ish x 128-ish
// output and a B*H*W-long reduction whose operands are each read ONCE: at 128 channels that is 64 FLOP per byte, and what
// bounds it on this chip is the ~10 B/clk/CU global-load path (L2 hits included), not HBM or MFMA - the 64 x 64 tiles of
// conv_wgrad_kernel<1> re-read both operands twice (32 FLOP per loaded byte -> 200 TFLOP/s measured).  Here a workgroup
// owns a 128 x 128 (n, c) tile (64 accumulator registers per lane) for a range of 64-pixel chunks, both operands arrive by
// LDS-DMA into a ring of 3 stages (same protocol as conv1x1_glds_kernel: counted vmcnt + one raw barrier per stage), rows are
// 128-byte pixel runs whose 16-byte chunk kc sits at chunk kc ^ (row & 7), fragments are plain ds_read_b128.
// Pixels past the plane and channels past the layer are fetched from a 16-byte page of zeros.
__device__ uint4 g_zero_page = {0u, 0u, 0u, 0u};
constexpr int kW2Threads = 256, kW2Ring = 3, kW2Px = 64;

template <bool SEG>
__device__ __forceinline__ void conv_wgrad1_glds_body(const ChanSegs &xs_, const uint16_t *__restrict__ dy, float *__restrict__ part,
                                                      int Cin, int Cout, int HW, int chunks_per_image, int total_chunks,
                                                      int chunks_per_split, int nct, int NP16, int CP16, int npairs, int nsplits,
                                                      const int bx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int OPB = 128 * 128, SB = 2 * OPB;                             // bytes per operand tile / per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pair, split;
    if ((nsplits & 7) == 0) {                                               // all tile pairs of a split on one XCD (L2 reuse)
        const int xcd = bx & 7, slot = bx >> 3;
        pair = slot % npairs; split = (slot / npairs) * 8 + xcd;
    } else {
        pair = bx % npairs; split = bx / npairs;
    }
    if (split >= nsplits) return;
    const int nt = pair / nct, ct = pair - nt * nct;
    const int n0 = nt * 128, c0 = ct * 128;
    const int g = lane >> 4, i16 = lane & 15;
    const int q0 = split * chunks_per_split, q1 = min(total_chunks, q0 + chunks_per_split);
    const int nstage = q1 - q0;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero_page);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const uint16_t *xflat = xs_.p[0];
    const uint16_t **xtab = reinterpret_cast<const uint16_t **>(lds + kW2Ring * SB);      // SEG: image-0 base per 8-channel group of this c tile
    if (SEG) {
        if (tid < 16) xtab[tid] = c0 + tid * 8 < Cin ? seg_addr(xs_, 0, c0 + tid * 8, HW) : nullptr;
        __syncthreads();
    }
    // batch stride (in channels) of the part a channel group lives in: second table
    int *xbs = reinterpret_cast<int *>(lds + kW2Ring * SB + 128);
    if (SEG) {
        if (tid < 16) {
            int sc = xs_.bs[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) if (k < xs_.n && c0 + tid * 8 >= xs_.start[k]) sc = xs_.bs[k];
            xbs[tid] = sc;
        }
        __syncthreads();
    }

    // this wave's 4 + 4 LDS-DMA pieces per stage: piece = 8 rows x 128 B; lane -> (row, physical chunk)
    int row_a[4], kc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        row_a[j] = row;
        kc[j] = ((lane & 7) ^ (row & 7)) << 3;                               // logical pixel offset of this lane's chunk
    }
    auto issue = [&](int s) {
        const int q = q0 + s;
        const int b = q / chunks_per_image, p0 = (q - b * chunks_per_image) * kW2Px;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (s % kW2Ring) * SB);
        const uint16_t *dyb = dy + (int64_t)b * Cout * HW + p0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pin = p0 + kc[j] < HW;
            const int n = n0 + row_a[j], c = c0 + row_a[j];
            glds16((pin && n < Cout) ? dyb + (int64_t)n * HW + kc[j] : zero, __builtin_amdgcn_readfirstlane(base + (wave * 4 + j) * 1024));
            const uint16_t *xp;
            if (SEG) {
                const int gr = min(row_a[j] >> 3, 15);
                xp = xtab[gr] + ((int64_t)b * xbs[gr] + (row_a[j] & 7)) * HW;
            } else {
                xp = xflat + ((int64_t)b * Cin + c) * HW;
            }
            glds16((pin && c < Cin) ? xp + p0 + kc[j] : zero, __builtin_amdgcn_readfirstlane(base + OPB + (wave * 4 + j) * 1024));
        }
    };
    f32x4v acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    if (nstage > 0) issue(0);
    if (nstage > 1) issue(1);
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char *ta = lds + (s % kW2Ring) * SB, *tb = ta + OPB;
        // all 20 fragments of the stage's two 32-pixel K steps are requested first, the copies of stage s + 2 are issued behind
        // them (their issue slots overlap the LDS latency), then 32 MFMAs run back to back - one wave per SIMD here, so nothing
        // else would hide a read -> wait -> MFMA chain
        bf16x8 af[2][2], bfr[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int row = wave * 32 + a * 16 + i16;
                af[ks][a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(ta + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4)));
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int row = b * 16 + i16;
                bfr[ks][b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(tb + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4)));
            }
        }
        if (s + 2 < nstage) issue(s + 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][a], bfr[ks][b], acc[a][b], 0, 0, 0);
    }
    // partial sums: part[split][n][c]; lane holds c = i16 (+ 16 b), n = 4 g + r (+ 16 a + 32 wave)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int c = c0 + b * 16 + i16;
        if (c >= CP16) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wave * 32 + a * 16 + 4 * g + r;
                if (n < NP16) part[((int64_t)split * NP16 + n) * CP16 + c] = acc[a][b][r];
            }
    }
}

template <bool SEG>
__global__ __launch_bounds__(kW2Threads) void conv_wgrad1_glds_kernel(const ChanSegs xs_, const uint16_t *__restrict__ dy,
                                                                      float *__restrict__ part, int Cin, int Cout, int HW,
                                                                      int chunks_per_image, int total_chunks, int chunks_per_split,
                                                                      int nct, int NP16, int CP16, int npairs, int nsplits) {
    conv_wgrad1_glds_body<SEG>(xs_, dy, part, Cin, Cout, HW, chunks_per_image, total_chunks, chunks_per_split, nct, NP16, CP16, npairs,
                               nsplits, blockIdx.x);
}

// Many 1x1 weight gradients (whole-tensor inputs) in ONE launch: grid = (largest problem, problems).  The small layers
// (128 -> 128 @ 40x40: 256 workgroups of four 64-pixel stages) are ~23 us on the device each, 53 of them per step: a fixed
// launch + ring-fill + drain cost per layer that one grid pays once.  table rows of 8 x int64 = {x, dy, ws, B, Cin, Cout,
// HW, splits | chunks per split << 32}.
__global__ __launch_bounds__(kW2Threads) void conv_wgrad1_group_kernel(const int64_t *__restrict__ table) {
    const int64_t *e = table + (int64_t)blockIdx.y * 8;
    const int B = (int)e[3], Cin = (int)e[4], Cout = (int)e[5], HW = (int)e[6];
    const int splits = (int)(e[7] & 0xffffffff), cps = (int)(e[7] >> 32);
    const int cpi = (HW + kW2Px - 1) / kW2Px;
    const int nnt = (Cout + 127) / 128, nct = (Cin + 127) / 128, npairs = nnt * nct;
    if ((int)blockIdx.x >= 8 * ((splits + 7) / 8) * npairs) return;
    ChanSegs xs_;
    xs_.p[0] = reinterpret_cast<const uint16_t *>(e[0]); xs_.start[0] = 0; xs_.start[1] = Cin; xs_.bs[0] = Cin; xs_.n = 1;
    conv_wgrad1_glds_body<false>(xs_, reinterpret_cast<const uint16_t *>(e[1]), reinterpret_cast<float *>(e[2]), Cin, Cout, HW, cpi,
                                 B * cpi, cps, nct, (Cout + 15) / 16 * 16, (Cin + 15) / 16 * 16, npairs, splits, blockIdx.x);
}

// wg_target: workgroups the problem should occupy - 256 (one per CU, one round: the load path of EVERY CU is needed) when it is
// launched on its own; a problem of a GROUPED launch (dfine_conv_wgrad1_group: 8 - 32 problems side by side) fills the chip
// with far fewer, and every split it does not use is a slab of fp32 partial sums that is not written and not read again by
// the deferred reduction (2.7 GB per D-FINE-m step with 256: the 27 128 x 128 layers alone 354 MB in 200 splits each)
static void wgrad1_plan(int B, int Cin, int Cout, int HW, int *splits, int *cps, int wg_target = 256) {
    const int cpi = (HW + kW2Px - 1) / kW2Px, total = B * cpi;
    const int pairs = ((Cout + 127) / 128) * ((Cin + 127) / 128);
    const int64_t bytes_per_split = (int64_t)((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * 4;
    int cap = (int)(24000000 / bytes_per_split);                            // fp32 partials (written once, read once) below ~24 MB
    if (cap < 8) cap = 8;
    int sp = wg_target / pairs;
    if (sp < 1) sp = 1;
    if (sp > cap) sp = cap;
    if (sp > total) sp = total;
    int c = (total + sp - 1) / sp;
    int spl = (total + c - 1) / c;
    if (spl >= 8 && (spl & 7)) {                                            // whole splits per XCD
        for (int u = c; u <= c * 4 / 3 + 1; ++u) {
            const int t = (total + u - 1) / u;
            if (t >= 8 && (t & 7) == 0) { c = u; spl = t; break; }
        }
    }
    *splits = spl; *cps = c;
}

// 64 outputs x 4 split lanes per block (a 128 x 128 layer has only 16 K outputs: one thread per output left 3/4 of
// the chip idle while every thread walked its ~256 partials one after the other).
// Blocks past the weight range (bias_blocks of them) sum the per-split bias partials part_b[split][NP16] -> db.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw,
                                                               int splits, int Cout, int Cin, int taps, int NP16, int CP16,
                                                               const float *__restrict__ part_b = nullptr,
                                                               float *__restrict__ db = nullptr, int w_blocks = 0) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (db && (int)blockIdx.x >= w_blocks) {
        const int n = ((int)blockIdx.x - w_blocks) * 64 + col;
        float s = 0.f;
        if (n < Cout)
            for (int k = q; k < splits; k += 4) s += part_b[(int64_t)k * NP16 + n];
        red[q][col] = s;
        __syncthreads();
        if (q == 0 && n < Cout) db[n] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        return;
    }
    const int64_t total = (int64_t)Cout * Cin * taps;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f;
    if (i < total) {
        const int t = (int)(i % taps);
        const int c = (int)((i / taps) % Cin);
        const int n = (int)(i / ((int64_t)taps * Cin));
        const float *src = part + ((int64_t)n * CP16 + c) * taps + t;
        const int64_t stride = (int64_t)NP16 * CP16 * taps;
        int k = q;
        for (; k + 4 < splits; k += 8) { s0 += src[(int64_t)k * stride]; s1 += src[(int64_t)(k + 4) * stride]; }
        for (; k < splits; k += 4) s0 += src[(int64_t)k * stride];
    }
    red[q][col] = s0 + s1;
    __syncthreads();
    if (q == 0 && i < total) dw[i] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of a token-stream linear layer: dW[n][k] = sum_m dY[m][n] * X[m][k] with ROW-major
// activations X [M, K], dY [M, N] (M = B*Lq = 15 744 rows, N, K <= 1024).  A GEMM with a tiny output and
// a very long reduction: as a single hipBLASLt `mm` it occupies <= 16 workgroups (91 us per call, 57 calls
// per D-FINE-m step).  Here the reduction is split over `splits` blocks per 64 x 64 output tile.
// The reduction index is the ROW index of both operands, so MFMA fragments (8 consecutive m per lane) are
// columns of the staged tiles: read from LDS with 16-bit loads (row pitch 66 elements -> the 4 row groups of
// a wave hit different banks).  LDS-read bound at ~20 % of the MFMA rate, far above what 2 GFLOP needs.
__device__ __forceinline__ void linear_wgrad_body(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                  float *__restrict__ part, float *__restrict__ db, int M, int N, int K,
                                                  int rows_per_split, int nct64, int NP16, int CP16, int tile, int split) {
    // Both MFMA operands are COLUMNS of row-major tiles (the reduction index m is the row): gfx950's LDS
    // transpose-read delivers them - lane (column i, group g) gets rows 4g..4g+3 (first read) and 16+4g..16+4g+3
    // (second read) of its column; A and B use the same row order, and a sum over m does not care about it.
    // Row pitch 80 elements = 160 B: the 8 rows a 32-lane half touches hit 8 different 32-byte bank groups.
    constexpr int PITCH = 80, ROWS = 64, NV = ROWS / 16;   // rows staged per barrier pair (two 32-row MFMA k-steps; 32: 74 -> 99 us, 128: 110 us at 1024 x 256)
    __shared__ __attribute__((aligned(16))) uint16_t s_dy[ROWS * PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t s_x[ROWS * PITCH];
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef short tr8 __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt64 = tile / nct64, ct64 = tile - nt64 * nct64;
    const int n0 = nt64 * 64, k0 = ct64 * 64;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    f32x4v acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v acc_b = f32x4v{0.f, 0.f, 0.f, 0.f};        // bias gradient: dY^T * ones (one extra MFMA per k-step)
    const bool want_db = db != nullptr && ct64 == 0;
    const bool wave_active = n0 + wave * 16 < N;
    // staging role: 128 threads per tile, thread -> 4 (row, 8-element chunk) vectors
    const int st_tile = tid >> 7, st_t = tid & 127;
    const uint16_t *st_src = st_tile == 0 ? dy + n0 : x + k0;
    const int st_ld = st_tile == 0 ? N : K;
    const int st_width = st_tile == 0 ? N - n0 : K - k0;
    const bool vec_ok = (st_ld & 7) == 0;                         // rows 16-byte aligned
    uint16_t *st_dst = st_tile == 0 ? s_dy : s_x;
    const int g = lane >> 4, i = lane & 15;
    const int tr_off = (4 * g + (i >> 2)) * PITCH + 4 * (i & 3);
    uint4 pf[NV];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            const int vi = st_t + 128 * h;                       // 0 .. 8 ROWS - 1
            const int row = vi >> 3, col = (vi & 7) * 8;
            const int m = m0 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < m_end && vec_ok && col + 8 <= st_width) v = *reinterpret_cast<const uint4 *>(st_src + (int64_t)m * st_ld + col);
            else if (m < m_end && col < st_width) {               // odd widths (132, 20, 4, 1 ...): element loads
                // (8-byte loads for the 132-wide box-head rows: 63 -> 43 us alone, but +0.33 ms per STEP - 27.34 -> 27.67 ms, three
                // builds alternated on one box - the grouped launch's slow 132-wide workgroups evidently pace the others usefully)
                uint16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int e = 0; e < min(8, st_width - col); ++e) tmp[e] = st_src[(int64_t)m * st_ld + col + e];
                v = *reinterpret_cast<uint4 *>(tmp);
            }
            pf[h] = v;
        }
    };
    fetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += ROWS) {
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            const int vi = st_t + 128 * h;
            *reinterpret_cast<uint4 *>(st_dst + (vi >> 3) * PITCH + (vi & 7) * 8) = pf[h];
        }
        __syncthreads();
        if (m0 + ROWS < m_end) fetch(m0 + ROWS);                   // next stage in flight during the MFMAs
        if (wave_active) {
#pragma unroll
            for (int ks = 0; ks < ROWS / 32; ++ks) {
                const uint16_t *pa = s_dy + ks * 32 * PITCH + tr_off + wave * 16;
                const tr4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)pa);
                const tr4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)(pa + 16 * PITCH));
                const bf16x8 a = __builtin_bit_cast(bf16x8, (tr8)__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
                if (want_db)
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        a, __builtin_bit_cast(bf16x8, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u)), acc_b, 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    if (k0 + ct * 16 >= K) continue;
                    const uint16_t *pb = s_x + ks * 32 * PITCH + tr_off + ct * 16;
                    const tr4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)pb);
                    const tr4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)(pb + 16 * PITCH));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, (tr8)__builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf, acc[ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int c = k0 + ct * 16 + (lane & 15);
        if (c >= CP16) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wave * 16 + 4 * (lane >> 4) + r;
            if (n < NP16) part[((int64_t)split * NP16 + n) * CP16 + c] = acc[ct][r];
        }
    }
    if (want_db && wave_active && (lane & 15) == 0) {          // per-split partial: no zero-fill, no atomics
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wave * 16 + 4 * (lane >> 4) + r;
            if (n < NP16) db[(int64_t)split * NP16 + n] = acc_b[r];
        }
    }
}

// The same product on 128 x 128 output tiles (wave = 64 x 64 = 4 x 4 MFMA tiles, 64 accumulator registers): half the L2 traffic
// and LDS fragment reads per MFMA of the 64 x 64 body - the body of the GROUPED launch (64 problems side by side fill the chip with
// 64 workgroups each: 801 -> 725 us for a decoder step's list, tools/linear_wgrad_group_bench.py, and 27.42 -> 27.27 ms per step on a
// same-box A/B of the two builds, tools/ab_trees.sh); a problem launched alone keeps the 64 x 64 body (256 workgroups: 74 us
// against 148).  Row pitch 144 elements = 288 B: the 8 rows a 32-lane half touches fall on 8 different 32-byte bank groups.
__device__ __forceinline__ void linear_wgrad_body128(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                     float *__restrict__ part, float *__restrict__ db, int M, int N, int K,
                                                     int rows_per_split, int nct128, int NP16, int CP16, int tile, int split) {
    constexpr int PITCH = 144, ROWS = 64, NV = ROWS / 8;
    __shared__ __attribute__((aligned(16))) uint16_t s_dy[ROWS * PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t s_x[ROWS * PITCH];
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef short tr8 __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int nt = tile / nct128, ct = tile - nt * nct128;
    const int n0 = nt * 128, k0 = ct * 128;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    f32x4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v acc_b[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc_b[a] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const bool want_db = db != nullptr && ct == 0 && wk == 0;
    const int wn0 = n0 + wn * 64, wk0 = k0 + wk * 64;
    const int na = min(4, max(0, (N - wn0 + 15) >> 4)), nb = min(4, max(0, (K - wk0 + 15) >> 4));
    const int st_tile = tid >> 7, st_t = tid & 127;
    const uint16_t *st_src = st_tile == 0 ? dy + n0 : x + k0;
    const int st_ld = st_tile == 0 ? N : K;
    const int st_width = st_tile == 0 ? N - n0 : K - k0;
    const bool vec_ok = (st_ld & 7) == 0;
    uint16_t *st_dst = st_tile == 0 ? s_dy : s_x;
    const int g = lane >> 4, i = lane & 15;
    const int tr_off = (4 * g + (i >> 2)) * PITCH + 4 * (i & 3);
    uint4 pf[NV];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            const int vi = st_t + 128 * h;
            const int row = vi >> 4, col = (vi & 15) * 8;
            const int m = m0 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < m_end && vec_ok && col + 8 <= st_width) v = *reinterpret_cast<const uint4 *>(st_src + (int64_t)m * st_ld + col);
            else if (m < m_end && col < st_width) {
                uint16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int e = 0; e < min(8, st_width - col); ++e) tmp[e] = st_src[(int64_t)m * st_ld + col + e];
                v = *reinterpret_cast<uint4 *>(tmp);
            }
            pf[h] = v;
        }
    };
    fetch(m_begin);
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
    for (int m0 = m_begin; m0 < m_end; m0 += ROWS) {
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            const int vi = st_t + 128 * h;
            *reinterpret_cast<uint4 *>(st_dst + (vi >> 4) * PITCH + (vi & 15) * 8) = pf[h];
        }
        __syncthreads();
        if (m0 + ROWS < m_end) fetch(m0 + ROWS);
        if (na > 0 && nb > 0) {
#pragma unroll
            for (int ks = 0; ks < ROWS / 32; ++ks) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const uint16_t *pa = s_dy + ks * 32 * PITCH + tr_off + wn * 64 + a * 16;
                    const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)pa);
                    const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)(pa + 16 * PITCH));
                    af[a] = __builtin_bit_cast(bf16x8, (tr8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint16_t *pb = s_x + ks * 32 * PITCH + tr_off + wk * 64 + b * 16;
                    const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)pb);
                    const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4 __attribute__((address_space(3))) *)(pb + 16 * PITCH));
                    bfr[b] = __builtin_bit_cast(bf16x8, (tr8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a >= na) break;
                    if (want_db) acc_b[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], ones, acc_b[a], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (b >= nb) break;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = wk0 + b * 16 + (lane & 15);
        if (c >= CP16) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = wn0 + a * 16 + 4 * (lane >> 4) + r;
                if (n < NP16) part[((int64_t)split * NP16 + n) * CP16 + c] = acc[a][b][r];
            }
    }
    if (want_db && (lane & 15) == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = wn0 + a * 16 + 4 * (lane >> 4) + r;
                if (n < NP16) db[(int64_t)split * NP16 + n] = acc_b[a][r];
            }
    }
}

__global__ __launch_bounds__(kConvThreads) void linear_wgrad_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                                    float *__restrict__ part, float *__restrict__ db, int M, int N,
                                                                    int K, int rows_per_split, int nct64, int NP16, int CP16) {
    linear_wgrad_body(x, dy, part, db, M, N, K, rows_per_split, nct64, NP16, CP16, blockIdx.x, blockIdx.y);
}

// Many linear weight gradients in ONE launch.  A token-stream weight gradient (M = 15 744 rows, 256 x 256 outputs) is ~1000
// short workgroups of three or four dependent memory round trips each: 16 us on the device for 2 us of traffic, 78 times per
// step.  The backward ops only REGISTER their (x, dY) pair; all pairs of a flush run as one grid = (largest problem, problems)
// whose latency chains overlap.  table rows of 8 x int64 = {x, dy, ws, M, N, K, rows per split, splits}; ws as in
// dfine_linear_wgrad_bf16 with dw == NULL (weight partials, then bias partials).
__global__ __launch_bounds__(kConvThreads) void linear_wgrad_group_kernel(const int64_t *__restrict__ table) {
    const int64_t *e = table + (int64_t)blockIdx.y * 8;
    const int M = (int)e[3], N = (int)e[4], K = (int)e[5], rows = (int)e[6], splits = (int)e[7];
    const int nnt = (N + 127) / 128, nct = (K + 127) / 128, ntiles = nnt * nct;
    if ((int)blockIdx.x >= ntiles * splits) return;
    const int split = blockIdx.x / ntiles, tile = blockIdx.x - split * ntiles;
    const int np16 = (N + 15) / 16 * 16, cp16 = (K + 15) / 16 * 16;
    float *ws = reinterpret_cast<float *>(e[2]);
    linear_wgrad_body128(reinterpret_cast<const uint16_t *>(e[0]), reinterpret_cast<const uint16_t *>(e[1]), ws,
                         ws + (int64_t)splits * np16 * cp16, M, N, K, rows, nct, np16, cp16, tile, split);
}

// Deferred split reduction of MANY weight gradients in one launch, accumulating into their final destination (the flat
// gradient buffer of the fused optimizer): table rows of 8 x int64 = {partials ptr, dst ptr, splits, Cout, Cin, taps, NP16, CP16}
// (a bias gradient is a row with Cin = taps = CP16 = 1); blockIdx.y = row, blockIdx.x strides over the row's outputs.
// Replaces one conv_wgrad_reduce_kernel launch per layer (179 per D-FINE-m step) + the per-parameter gradient tensors.
// One block sums the partials of kWrGroups(splits) consecutive 64-element groups of one table row; the grid is
// (largest group count of the launch, rows), blocks beyond a row's own count leave at once.  (A fixed 32 blocks per row
// left the small layers - up to 512 splits of a few KB - as 128 waves walking chains of dependent rounds: 1.44 TB/s over
// the step's 2.57 GB of partials.)
__host__ __device__ static inline int wr_groups(int splits) { return splits >= 128 ? 1 : 128 / (splits < 1 ? 1 : splits); }

__global__ __launch_bounds__(256) void multi_wgrad_reduce_kernel(const int64_t *__restrict__ table) {
    __shared__ float red[4][64];
    const int64_t *e = table + (int64_t)blockIdx.y * 8;
    const int splits = (int)e[2], Cout = (int)e[3], Cin = (int)e[4], taps = (int)e[5], NP16 = (int)e[6], CP16 = (int)e[7];
    const int64_t total = (int64_t)Cout * Cin * taps, stride = (int64_t)NP16 * CP16 * taps;
    const int per = wr_groups(splits);
    const float *part = reinterpret_cast<const float *>(e[0]);
    float *dst = reinterpret_cast<float *>(e[1]);
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (Cin == CP16) {
        // channel counts that are multiples of 16 (all but the stem / head layers): a split's partial sums are one contiguous run in
        // output order, so a lane takes FOUR consecutive outputs with 16-byte loads - a quarter of the load instructions for the
        // same bytes (the 4-byte form moved the step's 2.7 GB at 3.8 TB/s).  Four times the outputs per block: the blocks past
        // the row's own count (the grid is sized for 64 outputs per group) leave at once.
        const int64_t firstv = (int64_t)blockIdx.x * per * 256;
        if (firstv >= total) return;
        for (int g = 0; g < per; ++g) {
            const int64_t i = firstv + (int64_t)g * 256 + 4 * col;
            if (firstv + (int64_t)g * 256 >= total) break;       // uniform over the block
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                const float *src = part + i;
                int k = q;
                for (; k + 12 < splits; k += 16) {               // 4 x 16 bytes in flight per lane
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(src + (int64_t)(k + 4 * u) * stride);
                    s0.x += v[0].x + v[2].x; s0.y += v[0].y + v[2].y; s0.z += v[0].z + v[2].z; s0.w += v[0].w + v[2].w;
                    s1.x += v[1].x + v[3].x; s1.y += v[1].y + v[3].y; s1.z += v[1].z + v[3].z; s1.w += v[1].w + v[3].w;
                }
                for (; k < splits; k += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)k * stride);
                    s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                }
            }
            __shared__ float4 redv[4][64];
            redv[q][col] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
            __syncthreads();
            if (q == 0 && i < total) {
                float4 d = *reinterpret_cast<float4 *>(dst + i);
                const float4 a = redv[0][col], b = redv[1][col], c2 = redv[2][col], d2 = redv[3][col];
                d.x += (a.x + b.x) + (c2.x + d2.x); d.y += (a.y + b.y) + (c2.y + d2.y);
                d.z += (a.z + b.z) + (c2.z + d2.z); d.w += (a.w + b.w) + (c2.w + d2.w);
                *reinterpret_cast<float4 *>(dst + i) = d;
            }
            __syncthreads();
        }
        return;
    }
    const int64_t first = (int64_t)blockIdx.x * per * 64;
    if (first >= total) return;
    for (int g = 0; g < per; ++g) {
        const int64_t i = first + (int64_t)g * 64 + col;
        if (first + (int64_t)g * 64 >= total) break;             // uniform over the block
        float s0 = 0.f, s1 = 0.f;
        if (i < total) {
            const int t = (int)(i % taps);
            const int c = (int)((i / taps) % Cin);
            const int n = (int)(i / ((int64_t)taps * Cin));
            const float *src = part + ((int64_t)n * CP16 + c) * taps + t;
            int k = q;
            for (; k + 28 < splits; k += 32) {                    // 8 loads in flight per lane
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + 4 * u) * stride];
                s0 += (v[0] + v[2]) + (v[4] + v[6]);
                s1 += (v[1] + v[3]) + (v[5] + v[7]);
            }
            for (; k + 4 < splits; k += 8) { s0 += src[(int64_t)k * stride]; s1 += src[(int64_t)(k + 4) * stride]; }
            for (; k < splits; k += 4) s0 += src[(int64_t)k * stride];
        }
        red[q][col] = s0 + s1;
        __syncthreads();
        if (q == 0 && i < total) dst[i] += (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        __syncthreads();
    }
}

static void wgrad_plan(int B, int Cin, int Cout, int H, int W, int KS, int *R, int *strips, int *splits, int *ups) {
    *R = 160 / W < 1 ? 1 : 160 / W;
    if (*R > H) *R = H;
    *strips = (H + *R - 1) / *R;
    const int units = B * *strips;
    const int pairs = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    // enough blocks to fill the chip, but keep the fp32 partial-sum buffer (written once, read once by
    // the reduce kernel) below ~24 MB
    const int64_t bytes_per_split = (int64_t)((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * KS * KS * 4;
    int cap = (int)(24000000 / bytes_per_split);
    if (cap < 8) cap = 8;
    if (cap > 512) cap = 512;
    constexpr int target = 1024;
    int sp = target / (pairs * KS);
    if (sp < 1) sp = 1;
    if (sp > cap) sp = cap;
    if (sp > units) sp = units;
    *ups = (units + sp - 1) / sp;
    *splits = (units + *ups - 1) / *ups;
    // a multiple of 8 splits lets the kernel give every XCD whole splits (see conv_wgrad_kernel): search the nearby
    // units-per-split values for one that yields it without shrinking the grid by more than a quarter
    if (*splits >= 8 && (*splits & 7)) {
        for (int u = *ups; u <= *ups * 4 / 3 + 1; ++u) {
            const int spl = (units + u - 1) / u;
            if (spl >= 8 && (spl & 7) == 0) { *ups = u; *splits = spl; break; }
        }
    }
}

// the wave-specialised 3x3 kernel's shape range (the others run on conv_igemm_kernel: no accumulation)
static bool conv3x3_ws_ok(int B, int NP, int KP, int H, int W) {
    if (W < 1 || W > 160 || H < 1) return false;
    int R = 160 / W;
    if (R > H) R = H;
    const int strips = (H + R - 1) / R;
    const size_t slab_bytes = (size_t)(R + 2) * (W + 2) * 64;
    int kc = KP >= 256 ? 4 : (KP >= 64 ? 2 : 1);
    while (kc > 1 && (slab_bytes * kc > 65536 || KP < 32 * kc)) kc >>= 1;
    const int vec = (W % 8 == 0) ? 8 : (W % 4 == 0 ? 4 : 2);
    const bool wide = (NP % 128 == 0) && ((int64_t)B * strips * (NP / 128) >= 512);
    const size_t ws_ep = (size_t)4 * 16 * (wide ? 2 : 1) * (16 * kMaxPixTiles + 8) * 2;
    const int kc_ws = kc > 2 ? 2 : kc;
    const size_t ws_slab = (size_t)(R + 2) * ws_pitch(W) * 64;
    const int ws_items = 16 * (R + 2) * (W / vec) * kc_ws;
    return NP % 64 == 0 && vec >= 4 && 2 * ws_slab * kc_ws + ws_ep <= 160 * 1024 && ws_items <= (vec == 8 ? 8 : 12) * 256;
}

static int launch_conv(const uint16_t *x, const uint16_t *w2, uint16_t *y, int B, int Cin, int Cout, int NP, int KP,
                       int H, int W, int KS, hipStream_t st, int accum = 0, const EpiAffine epi = EpiAffine{nullptr, nullptr, nullptr, 0}) {
    if (KS == 3 && conv3x3_rows32_ok(NP, KP, H, W))
        return epi.scale ? DFINE_E_BADARG : conv3x3_rows32_launch(x, w2, y, B, Cin, Cout, NP, KP, H, W, accum, st);
    // strip height: as many rows as fit in 160 pixels
    int R = 160 / W;
    if (R < 1) return DFINE_E_BADARG;
    if (R > H) R = H;
    const int strips = (H + R - 1) / R;
    const int pad = KS / 2;
    const size_t slab_bytes = (size_t)(R + 2 * pad) * (W + 2 * pad) * 64;
    // slabs per stage: deep input-channel counts amortise the stage latency over 64 / 128 channels
    int kc = KP >= 256 ? 4 : (KP >= 64 ? 2 : 1);
    while (kc > 1 && (slab_bytes * kc > 65536 || KP < 32 * kc)) kc >>= 1;
    const size_t ldsb = slab_bytes * kc;
    const int vec = (W % 8 == 0) ? 8 : (W % 4 == 0 ? 4 : 2);
    const int nblk64 = (NP + 63) / 64;
    const bool wide = (NP % 128 == 0) && ((int64_t)B * strips * (NP / 128) >= 512);
    const size_t ws_ep = (size_t)4 * 16 * (wide ? 2 : 1) * (16 * kMaxPixTiles + 8) * 2;
    int kc_ws = kc > 2 ? 2 : kc;                               // two buffers: the stage overhead is already hidden
    const size_t ws_slab = (size_t)(R + 2 * pad) * ws_pitch(W) * 64;
    if (KS == 3 && conv3x3_ws_ok(B, NP, KP, H, W)) {
        // wave-specialised persistent kernel: 128-channel blocks when they fill the chip, 64-channel blocks otherwise
        const int ntn = wide ? 2 : 1;
        const int nblk = NP / (64 * ntn);
        const int units = B * strips * nblk;
        static const int cus = [] { hipDeviceProp_t p; return hipGetDeviceProperties(&p, 0) == hipSuccess ? p.multiProcessorCount : 256; }();
        const int upx = (units + 7) / 8;
        const int wpx = upx < cus / 8 ? upx : cus / 8;
        static bool attr_ws = false;
        if (!attr_ws) {
            hipError_t e = hipSuccess, r;
#define DFINE_WS_ATTR(N, V, K) \
    if ((r = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_ws_kernel<N, V, K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) e = r; \
    if ((r = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_ws_kernel<N, V, K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) e = r;
            DFINE_WS_ATTR(1, 8, 1) DFINE_WS_ATTR(1, 8, 2) DFINE_WS_ATTR(2, 8, 1) DFINE_WS_ATTR(2, 8, 2)
            DFINE_WS_ATTR(1, 4, 1) DFINE_WS_ATTR(1, 4, 2) DFINE_WS_ATTR(2, 4, 1) DFINE_WS_ATTR(2, 4, 2)
#undef DFINE_WS_ATTR
            if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
            attr_ws = true;
        }
#define DFINE_WSE(N, V, K, E) \
    hipLaunchKernelGGL((conv3x3_ws_kernel<N, V, K, E>), dim3(8 * wpx), dim3(kWsThreads), 2 * ws_slab * kc_ws + ws_ep, st, x, w2, y, Cin, Cout, NP, KP, H, W, R, strips, nblk, units, accum, epi)
#define DFINE_WS(N, V, K) { if (epi.scale) DFINE_WSE(N, V, K, true); else DFINE_WSE(N, V, K, false); }
#define DFINE_WS_K(N, V) { if (kc_ws == 2) DFINE_WS(N, V, 2) else DFINE_WS(N, V, 1) }
#define DFINE_WS_V(N) { if (vec == 8) DFINE_WS_K(N, 8) else DFINE_WS_K(N, 4) }
        if (ntn == 2) DFINE_WS_V(2) else DFINE_WS_V(1)
#undef DFINE_WS_V
#undef DFINE_WS_K
#undef DFINE_WS
#undef DFINE_WSE
        return check_launch();
    }
    if (epi.scale) return DFINE_E_BADARG;                // (the first-generation kernel has no epilogue)
    dim3 grid(B * strips, wide ? NP / 128 : nblk64);
#define DFINE_CONV(KSS, NTNN, VECC, KCC)                                                                   \
    hipLaunchKernelGGL((conv_igemm_kernel<KSS, NTNN, VECC, KCC>), grid, dim3(kConvThreads), ldsb, st, x, w2, y, Cin, \
                       Cout, NP, KP, H, W, R, strips, accum)
#define DFINE_CONV_K(KSS, NTNN, VECC)                                                                 \
    { if (kc == 4) DFINE_CONV(KSS, NTNN, VECC, 4); else if (kc == 2) DFINE_CONV(KSS, NTNN, VECC, 2); else DFINE_CONV(KSS, NTNN, VECC, 1); }
#define DFINE_CONV_V(KSS, NTNN)                                                                       \
    { if (vec == 8) DFINE_CONV_K(KSS, NTNN, 8) else if (vec == 4) DFINE_CONV_K(KSS, NTNN, 4) else DFINE_CONV_K(KSS, NTNN, 2) }
    if (KS == 3) { if (wide) DFINE_CONV_V(3, 2) else DFINE_CONV_V(3, 1) }
    else return DFINE_E_BADARG;                       // 1x1 layers: conv1x1_tr_kernel (launch_conv1x1)
#undef DFINE_CONV_V
#undef DFINE_CONV_K
#undef DFINE_CONV
    return check_launch();
}

}  // namespace dfine

using namespace dfine;

static int launch_wgrad1(const ChanSegs &xs_, const void *dy, float *dw, float *ws, int B, int Cin, int Cout, int HW, hipStream_t st) {
    int splits, cps;
    wgrad1_plan(B, Cin, Cout, HW, &splits, &cps);
    const int cpi = (HW + kW2Px - 1) / kW2Px;
    const int nnt = (Cout + 127) / 128, nct = (Cin + 127) / 128, npairs = nnt * nct;
    const int np16 = (Cout + 15) / 16 * 16, cp16 = (Cin + 15) / 16 * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad1_glds_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kW2Ring * 2 * 128 * 128);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad1_glds_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kW2Ring * 2 * 128 * 128 + 256);
        if (e != hipSuccess || e2 != hipSuccess) { set_last_error(e != hipSuccess ? e : e2); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
    if (xs_.n > 1 || xs_.bs[0] != Cin)
        hipLaunchKernelGGL(conv_wgrad1_glds_kernel<true>, dim3(splits * npairs), dim3(kW2Threads), (size_t)kW2Ring * 2 * 128 * 128 + 256, st, xs_,
                           (const uint16_t *)dy, ws, Cin, Cout, HW, cpi, B * cpi, cps, nct, np16, cp16, npairs, splits);
    else
        hipLaunchKernelGGL(conv_wgrad1_glds_kernel<false>, dim3(splits * npairs), dim3(kW2Threads), (size_t)kW2Ring * 2 * 128 * 128, st, xs_,
                           (const uint16_t *)dy, ws, Cin, Cout, HW, cpi, B * cpi, cps, nct, np16, cp16, npairs, splits);
    if (int e = check_launch()) return e;
    if (!dw) return DFINE_OK;                                // partials only: reduced later by dfine_multi_wgrad_reduce
    const int64_t total = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((int)((total + 63) / 64)), dim3(256), 0, st, ws, dw, splits, Cout, Cin, 1,
                       np16, cp16);
    return check_launch();
}

static bool make_segs(ChanSegs *sg, const void *const *parts, const int *channels, const int *bstrides, int n, int C) {
    if (!parts || !channels || n < 1 || n > 8) return false;
    int c = 0;
    for (int k = 0; k < n; ++k) {
        if (!parts[k] || channels[k] < 1 || (bstrides && bstrides[k] < channels[k])) return false;
        sg->p[k] = (const uint16_t *)parts[k]; sg->start[k] = c; sg->bs[k] = bstrides ? bstrides[k] : channels[k]; c += channels[k];
    }
    sg->start[n] = c; sg->n = n;
    return c == C;
}

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

// Packs n_entries weight tensors with one launch.  table: device int64 [n_entries][8] =
// {w ptr (fp32 [Cout,Cin,KS,KS]), w2 ptr, Cout, Cin, KS, NP, KP, dgrad} with NP / KP as in dfine_conv_packed_elems.
int dfine_conv_pack_weights_multi(const void *table, int n_entries, void *stream) {
    if (n_entries == 0) return DFINE_OK;
    if (!table || n_entries < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(conv_pack_weights_multi_kernel, dim3(64, n_entries), dim3(256), 0, (hipStream_t)stream,
                       (const int64_t *)table);
    return check_launch();
}

// y[B, Cout, H, W] = conv(x[B, Cin, H, W], packed weights), stride 1, padding KS/2, bf16.
// `w2` comes from dfine_conv_pack_weights(dgrad = 0) - or (dgrad = 1) with Cin/Cout exchanged by the
// caller, which makes this the data gradient dX = conv(dY, flipped-transposed weights).
// One-shot request consumed by the next dfine_conv_fwd_bf16 / dfine_conv1x1_seg_fwd_bf16 of the calling thread:
//     y[n] = lab[0] * act(scale[n] * conv[n] + shift[n]) + lab[1]      (act: 0 none, 1 ReLU, 2 SiLU; lab: 2 floats or NULL)
// applied to the fp32 accumulators in the kernel's store phase - a conv -> eval-mode BatchNorm (or deployed bias) -> act
// [-> learnable affine] unit as one launch.  Only the shapes dfine_conv_affine_supported() accepts are served; a launch that
// cannot returns DFINE_E_BADARG and drops the request.
static thread_local EpiAffine g_conv_epi = {nullptr, nullptr, nullptr, 0};
int dfine_conv_affine_once(const float *scale, const float *shift, const float *lab, int act) {
    g_conv_epi = EpiAffine{nullptr, nullptr, nullptr, 0};
    if (!scale) return DFINE_OK;                          // (scale == NULL withdraws a pending request)
    if (!shift || act < 0 || act > 2) return DFINE_E_BADARG;
    g_conv_epi = EpiAffine{scale, shift, lab, act};
    return DFINE_OK;
}

int dfine_conv_affine_supported(int B, int Cin, int Cout, int H, int W, int KS) {
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || Cin % 2) return 0;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    if (KS == 1) return ((H * W) % 2 == 0 && conv1x1_glds_ok(Cin, KP, H * W)) ? 1 : 0;
    if (KS == 3) return (W % 2 == 0 && W <= 160 && !conv3x3_rows32_ok(NP, KP, H, W) && conv3x3_ws_ok(B, NP, KP, H, W)) ? 1 : 0;
    return 0;
}

int dfine_conv_fwd_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int H, int W, int KS,
                        void *stream) {
    const EpiAffine epi = g_conv_epi;
    g_conv_epi = EpiAffine{nullptr, nullptr, nullptr, 0};
    if (B == 0) return DFINE_OK;
    if (!x || !w2 || !y || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (KS != 1 && KS != 3)) return DFINE_E_BADARG;
    if (Cin % 2) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    if (KS == 1) {             // no spatial structure: flattened planes, LDS transpose-read kernel
        if ((H * W) % 2) return DFINE_E_BADARG;
        return launch_conv1x1((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, H * W,
                              (hipStream_t)stream, nullptr, nullptr, 0, 0, 0, epi);
    }
    const int h = H, w = W;
    if (w % 2 || w > 160) return DFINE_E_BADARG;
    return launch_conv((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, h, w, KS,
                       (hipStream_t)stream, 0, epi);
}

// y += conv(x): a data gradient added onto the one already in y - the sum autograd forms for a map with two consumers
// (HG_Block: layer i output -> layer i + 1 and the aggregation, ref hgnetv2.py:265-274) without the element-wise add pass.
// Only on the shapes dfine_conv_epilogue_supported() accepts (the LDS-DMA 1x1 kernel / the wave-specialised 3x3 kernel);
// DFINE_E_BADARG otherwise, the caller then adds separately.
int dfine_conv_epilogue_supported(int B, int Cin, int Cout, int H, int W, int KS) {
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || Cin % 2) return 0;
    const int KP = (Cin + 31) / 32 * 32;
    if (KS == 1) return conv1x1_glds_ok(Cin, KP, H * W) ? 1 : 0;
    // (3x3: the wave-specialised kernel, or - output channels not a multiple of 64: stage 1 of the backbone - the first-generation
    // kernel, whose per-element store reads the old value first)
    if (KS == 3) return (W % 2 == 0 && W <= 160) ? 1 : 0;
    return 0;
}

int dfine_conv_accum_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int H, int W, int KS, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !w2 || !y || !dfine_conv_epilogue_supported(B, Cin, Cout, H, W, KS)) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    if (KS == 1)
        return launch_conv1x1((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, H * W,
                              (hipStream_t)stream, nullptr, nullptr, 1);
    return launch_conv((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, H, W, KS,
                       (hipStream_t)stream, 1);
}

// y[B, Cout, H*W] += conv1x1(x[B, Cin, H*W]) (bf16 accumulate-into: the second data gradient of a RepVGG unit adds onto
// the first instead of a separate add pass).  DFINE_E_BADARG when the shape is outside the LDS-DMA kernel's range
// (H*W % 8, Cin % 4): the caller then adds separately.
int dfine_conv1x1_accum_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int HW, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !w2 || !y || Cin < 1 || Cout < 1 || HW < 1 || Cin % 2) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    return launch_conv1x1((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, HW,
                          (hipStream_t)stream, nullptr, nullptr, 1);
}

// y[b] = conv1x1(x[b], W_b) with ONE WEIGHT SET PER IMAGE: w2 [B][NP][KP] packed like dfine_conv_pack_weights(KS = 1) per
// image (NP = Cout rounded up to 16, KP = Cin rounded up to 32, k in tr_slab_channel order).  The mask-logit contraction
// einsum("bqc,bchw->bqhw") of the segmentation head (ref dfine_decoder.py:925-932: Cout = queries, Cin = mask_dim) and its
// gradient with respect to the mask features (weights = the embeddings transposed).  (H*W) % 8 == 0, Cin % 4 == 0.
int dfine_conv1x1_bw_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int HW, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !w2 || !y || Cin < 4 || Cout < 1 || HW < 1 || Cin % 4 || HW % 8) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    return launch_conv1x1((const uint16_t *)x, (const uint16_t *)w2, (uint16_t *)y, B, Cin, Cout, NP, KP, HW, (hipStream_t)stream,
                          nullptr, nullptr, 0, (int64_t)NP * KP);
}

// Weight gradient of the same convolution.  x [B,Cin,H,W], dy [B,Cout,H,W] bf16 -> dw [Cout,Cin,KS,KS]
// f32 (overwritten).  ws: dfine_conv_wgrad_ws_floats(...) floats.  KS = 3 needs W % 8 == 0 and
// W <= 160; KS = 1 needs (H*W) % 8 == 0.
static bool wgrad1_v2(int KS, int HW) {
    static const int env = [] { const char *e = getenv("DFINE_WGRAD1_GLDS"); return e ? atoi(e) : 1; }();
    return env && KS == 1 && HW % 8 == 0;
}

int64_t dfine_conv_wgrad_ws_floats(int B, int Cin, int Cout, int H, int W, int KS) {
    if (wgrad1_v2(KS, H * W)) {
        int splits, cps;
        wgrad1_plan(B, Cin, Cout, H * W, &splits, &cps);
        return (int64_t)splits * ((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16);
    }
    if (KS == 3) {
        const int s3 = wgrad3_rows_splits(B, Cin, Cout, H, W);
        if (s3 > 0) return (int64_t)s3 * ((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * 9;
    }
    int h = H, w = W;
    if (KS == 1) { const int hw = H * W; w = 160; while (w > 8 && (hw % w || w % 8)) --w; h = hw / w; }
    int R, strips, splits, ups;
    wgrad_plan(B, Cin, Cout, h, w, KS, &R, &strips, &splits, &ups);
    return (int64_t)splits * ((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * KS * KS;
}

int dfine_conv_wgrad_bf16(const void *x, const void *dy, float *dw, float *ws, int B, int Cin, int Cout, int H,
                          int W, int KS, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !dy || !ws || Cin < 1 || Cout < 1 || (KS != 1 && KS != 3)) return DFINE_E_BADARG;
    if (wgrad1_v2(KS, H * W)) return launch_wgrad1(one_seg(x, Cin), dy, dw, ws, B, Cin, Cout, H * W, (hipStream_t)stream);
    if (KS == 3) {
        const int s3 = wgrad3_rows_splits(B, Cin, Cout, H, W);
        if (s3 > 0) {
            if (int e = wgrad3_rows_launch(x, dy, ws, B, Cin, Cout, H, W, (hipStream_t)stream)) return e;
            if (!dw) return DFINE_OK;                            // partials only
            const int64_t total = (int64_t)Cout * Cin * 9;
            hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((int)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, ws, dw, s3,
                               Cout, Cin, 9, (Cout + 15) / 16 * 16, (Cin + 15) / 16 * 16);
            return check_launch();
        }
    }
    int h = H, w = W;
    if (KS == 1) {
        const int hw = H * W;
        w = 160;
        while (w > 8 && (hw % w || w % 8)) --w;
        if (hw % w || w % 8) return DFINE_E_BADARG;
        h = hw / w;
    }
    if (w % 8 || w > 160) return DFINE_E_BADARG;
    int R, strips, splits, ups;
    wgrad_plan(B, Cin, Cout, h, w, KS, &R, &strips, &splits, &ups);
    const int nnt64 = (Cout + 63) / 64, nct64 = (Cin + 63) / 64;
    const int lpad = KS == 3 ? 8 : 0;
    const int np16 = (Cout + 15) / 16 * 16, cp16 = (Cin + 15) / 16 * 16;
    // LDS elements per channel.  ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...),
    // each mixing two of the wave's K groups (pixel offset +16 B): with a channel stride of 16 B (mod 256 B) lanes
    // (channel 12, group 0) and (channel 11, group 1) share a 16-byte slot and every read costs 2x (42 % conflict
    // cycles, profiles/r01_conv_pmc.txt); a stride of 32 B (mod 256 B) puts group 0 on even and group 1 on odd slots.
    // The 3x3 kernel keeps 16 B: its two extra ds_read_b32 per fragment would be 4-way conflicted at 32 B.
    const int cs = (R * (w + 2 * lpad) + 127) / 128 * 128 + (KS == 1 ? 16 : 8);
    const size_t ldsb = (size_t)64 * cs * 2;
    hipStream_t st = (hipStream_t)stream;
    const int npairs = nnt64 * nct64;
    dim3 grid(splits * npairs, 1, KS);
    if (KS == 3) {
        static bool attr_set = false;       // once: not a stream operation, keep it out of graph capture
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_kernel<3>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
            attr_set = true;
        }
        hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(kConvThreads), ldsb, st, (const uint16_t *)x, (const uint16_t *)dy, ws,
                           Cin, Cout, h, w, R, strips, B * strips, ups, nct64, np16, cp16, cs, npairs, splits);
    } else {
        hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(kConvThreads), ldsb, st, (const uint16_t *)x, (const uint16_t *)dy, ws,
                           Cin, Cout, h, w, R, strips, B * strips, ups, nct64, np16, cp16, cs, npairs, splits);
    }
    if (int e = check_launch()) return e;
    if (!dw) return DFINE_OK;                                // partials only
    const int64_t total = (int64_t)Cout * Cin * KS * KS;
    const int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, dw, splits, Cout, Cin, KS * KS,
                       np16, cp16);
    return check_launch();
}

// 1x1 convolution whose input and / or output is a channel-wise concatenation kept as separate tensors: x_parts / y_parts are
// HOST arrays of n device pointers (part k = [B, channels[k], H, W] bf16 with batch stride bstrides[k] * H * W elements - the
// channel count, or more for a channel slice of a wider tensor; NULL = contiguous), sum(channels) = Cin / Cout.  (H*W) % 8 == 0.
int dfine_conv1x1_seg_fwd_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x, const void *w2,
                               void *const *y_parts, const int *y_channels, const int *y_bstrides, int n_y, int B, int Cin, int Cout,
                               int H, int W, void *stream) {
    const EpiAffine epi = g_conv_epi;                     // (dfine_conv_affine_once)
    g_conv_epi = EpiAffine{nullptr, nullptr, nullptr, 0};
    if (B == 0) return DFINE_OK;
    ChanSegs xs_, ys_;
    if (!w2 || !make_segs(&xs_, x_parts, x_channels, x_bstrides, n_x, Cin) ||
        !make_segs(&ys_, (const void *const *)y_parts, y_channels, y_bstrides, n_y, Cout))
        return DFINE_E_BADARG;
    if ((H * W) % 8 || Cin % 2) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    return launch_conv1x1(nullptr, (const uint16_t *)w2, nullptr, B, Cin, Cout, NP, KP, H * W, (hipStream_t)stream, &xs_, &ys_, 0, 0, 0, epi);
}

// y_parts += the same convolution: a data gradient added onto the channel slice of a wider gradient map that already holds the
// other consumers' terms (RepNCSPELAN4: cv1's output feeds cv4 whole and the CSP branch through its upper half, ref
// hybrid_encoder.py:196-206) - no data-gradient tensor of its own, no element-wise add, no zero-filled slice gradient.
int dfine_conv1x1_seg_accum_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x, const void *w2,
                                 void *const *y_parts, const int *y_channels, const int *y_bstrides, int n_y, int B, int Cin, int Cout,
                                 int H, int W, void *stream) {
    if (B == 0) return DFINE_OK;
    ChanSegs xs_, ys_;
    if (!w2 || !make_segs(&xs_, x_parts, x_channels, x_bstrides, n_x, Cin) ||
        !make_segs(&ys_, (const void *const *)y_parts, y_channels, y_bstrides, n_y, Cout))
        return DFINE_E_BADARG;
    if ((H * W) % 8 || Cin % 2) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    if (!conv1x1_glds_ok(Cin, KP, H * W)) return DFINE_E_BADARG;           // only the LDS-DMA kernel accumulates
    return launch_conv1x1(nullptr, (const uint16_t *)w2, nullptr, B, Cin, Cout, NP, KP, H * W, (hipStream_t)stream, &xs_, &ys_, 1);
}

// ... onto SOME of the output parts (bit k of accum_parts), the others are overwritten: the data gradient of HG_Block's aggregation
// written into one tensor per concatenated input, where the block input's tensor already holds the gradient of the residual
// connection (ref hgnetv2.py:265-275).
int dfine_conv1x1_seg_accum_parts_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x, const void *w2,
                                       void *const *y_parts, const int *y_channels, const int *y_bstrides, int n_y, unsigned accum_parts,
                                       int B, int Cin, int Cout, int H, int W, void *stream) {
    if (B == 0) return DFINE_OK;
    ChanSegs xs_, ys_;
    if (!w2 || !make_segs(&xs_, x_parts, x_channels, x_bstrides, n_x, Cin) ||
        !make_segs(&ys_, (const void *const *)y_parts, y_channels, y_bstrides, n_y, Cout))
        return DFINE_E_BADARG;
    if ((H * W) % 8 || Cin % 2 || (n_y < 32 && (accum_parts >> n_y))) return DFINE_E_BADARG;
    const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
    if (!conv1x1_glds_ok(Cin, KP, H * W)) return DFINE_E_BADARG;
    return launch_conv1x1(nullptr, (const uint16_t *)w2, nullptr, B, Cin, Cout, NP, KP, H * W, (hipStream_t)stream, &xs_, &ys_, 0, 0,
                          accum_parts);
}

// weight gradient of the same: dw [Cout, Cin] f32 (overwritten), ws: dfine_conv_wgrad_ws_floats(B, Cin, Cout, H, W, 1) floats
int dfine_conv1x1_seg_wgrad_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x, const void *dy,
                                 float *dw, float *ws,
                                 int B, int Cin, int Cout, int H, int W, void *stream) {
    if (B == 0) return DFINE_OK;
    ChanSegs xs_;
    if (!dy || !ws || !make_segs(&xs_, x_parts, x_channels, x_bstrides, n_x, Cin) || (H * W) % 8) return DFINE_E_BADARG;
    return launch_wgrad1(xs_, dy, dw, ws, B, Cin, Cout, H * W, (hipStream_t)stream);
}

static void linear_wgrad_plan(int M, int N, int K, int *splits, int *rows) {
    const int pairs = ((N + 63) / 64) * ((K + 63) / 64);
    constexpr int wgs = 256;     // (1024: +0.25 ms per step - 4x the partial sums)
    int sp = wgs / pairs;
    if (sp < 1) sp = 1;
    if (sp > 128) sp = 128;
    int r = ((M + sp - 1) / sp + 63) / 64 * 64;
    if (r < 32) r = 32;
    *rows = r;
    *splits = (M + r - 1) / r;
}

int64_t dfine_linear_wgrad_ws_floats(int M, int N, int K) {
    int splits, rows;
    linear_wgrad_plan(M, N, K, &splits, &rows);
    const int64_t np16 = (N + 15) / 16 * 16;
    return (int64_t)splits * np16 * ((K + 15) / 16 * 16) + (int64_t)splits * np16;     // weight partials + bias partials
}

// dw [N, K] f32 (overwritten) = dy[M, N]^T x[M, K]; db [N] f32 (overwritten, may be NULL) = column sums of dy; x, dy row-major bf16 (16-byte loads when the row length is a
// multiple of 8, element loads otherwise).
int dfine_linear_wgrad_bf16(const void *x, const void *dy, float *dw, float *db, float *ws, int M, int N, int K,
                            void *stream) {
    if (M == 0) return DFINE_OK;
    if (!x || !dy || !ws || M < 1 || N < 1 || K < 1) return DFINE_E_BADARG;
    int splits, rows;
    linear_wgrad_plan(M, N, K, &splits, &rows);
    const int nnt64 = (N + 63) / 64, nct64 = (K + 63) / 64;
    const int np16 = (N + 15) / 16 * 16, cp16 = (K + 15) / 16 * 16;
    hipStream_t st = (hipStream_t)stream;
    float *part_b = (db || !dw) ? ws + (int64_t)splits * np16 * cp16 : nullptr;     // dw == NULL: partials (weights AND bias) only
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(nnt64 * nct64, splits), dim3(kConvThreads), 0, st, (const uint16_t *)x,
                       (const uint16_t *)dy, ws, part_b, M, N, K, rows, nct64, np16, cp16);
    if (int e = check_launch()) return e;
    if (!dw) return DFINE_OK;
    const int64_t total = (int64_t)N * K;
    const int blocks = (int)((total + 63) / 64);
    const int bias_blocks = db ? (N + 63) / 64 : 0;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(blocks + bias_blocks), dim3(256), 0, st, ws, dw, splits, N, K, 1, np16,
                       cp16, (const float *)part_b, db, blocks);
    return check_launch();
}

// Number of splits (partial-sum slabs) the weight-gradient entry points above write into `ws` for these shapes.
int dfine_conv_wgrad_splits(int B, int Cin, int Cout, int H, int W, int KS) {
    if (wgrad1_v2(KS, H * W)) {
        int splits, cps;
        wgrad1_plan(B, Cin, Cout, H * W, &splits, &cps);
        return splits;
    }
    if (KS == 3) {
        const int s3 = wgrad3_rows_splits(B, Cin, Cout, H, W);
        if (s3 > 0) return s3;
    }
    int h = H, w = W;
    if (KS == 1) { const int hw = H * W; w = 160; while (w > 8 && (hw % w || w % 8)) --w; h = hw / w; }
    int R, strips, splits, ups;
    wgrad_plan(B, Cin, Cout, h, w, KS, &R, &strips, &splits, &ups);
    return splits;
}

int dfine_linear_wgrad_splits(int M, int N, int K) {
    int splits, rows;
    linear_wgrad_plan(M, N, K, &splits, &rows);
    return splits;
}

// Fills one 8 x int64 table row of dfine_linear_wgrad_group for a problem and returns the workgroups it needs.
int dfine_linear_wgrad_group_row(const void *x, const void *dy, float *ws, int M, int N, int K, int64_t *row) {
    if (!x || !dy || !ws || !row || M < 1 || N < 1 || K < 1) return DFINE_E_BADARG;
    int splits, rows;
    linear_wgrad_plan(M, N, K, &splits, &rows);
    row[0] = (int64_t)x; row[1] = (int64_t)dy; row[2] = (int64_t)ws; row[3] = M; row[4] = N; row[5] = K; row[6] = rows; row[7] = splits;
    return ((N + 127) / 128) * ((K + 127) / 128) * splits;
}

// 1x1 weight gradients, grouped: row helper (returns the workgroup count, < 0 when this shape does not take the LDS-DMA kernel).
static int wgrad1_group_target() {
    // measured (D-FINE-m step, ms median): 256 -> 33.29 / 33.41, 128 -> 33.37, 64 -> 33.51 / 33.58, 32 -> 33.91: the kernels gain more
    // from the parallelism of many splits than the step loses to their partial sums (2.7 -> 1.6 GB at 64) - the default keeps 256
    // (re-measured at the end of round 5, 29.0 ms steps: 256 -> 29.04 / 29.03, 128 -> 29.02 / 29.01, 64 -> 29.18: 128 for half the partial sums)
    return 128;
}

static void wgrad1_group_plan(int B, int Cin, int Cout, int HW, int *splits, int *cps) {
    wgrad1_plan(B, Cin, Cout, HW, splits, cps, wgrad1_group_target());
}

// Splits (partial-sum slabs) and workspace floats of a problem of the GROUPED 1x1 weight-gradient launch.
int dfine_conv_wgrad1_group_splits(int B, int Cin, int Cout, int HW) {
    int splits, cps;
    wgrad1_group_plan(B, Cin, Cout, HW, &splits, &cps);
    return splits;
}

int64_t dfine_conv_wgrad1_group_ws_floats(int B, int Cin, int Cout, int HW) {
    return (int64_t)dfine_conv_wgrad1_group_splits(B, Cin, Cout, HW) * ((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16);
}

int dfine_conv_wgrad1_group_row(const void *x, const void *dy, float *ws, int B, int Cin, int Cout, int HW, int64_t *row) {
    if (!x || !dy || !ws || !row || B < 1 || Cin < 1 || Cout < 1 || !wgrad1_v2(1, HW)) return DFINE_E_BADARG;
    int splits, cps;
    constexpr int tile = 128;
    wgrad1_group_plan(B, Cin, Cout, HW, &splits, &cps);
    row[0] = (int64_t)x; row[1] = (int64_t)dy; row[2] = (int64_t)ws; row[3] = B; row[4] = Cin; row[5] = Cout; row[6] = HW;
    row[7] = (int64_t)splits | ((int64_t)cps << 32);
    return 8 * ((splits + 7) / 8) * ((Cout + tile - 1) / tile) * ((Cin + tile - 1) / tile);
}

int dfine_conv_wgrad1_group(const void *table, int n_problems, int max_blocks, void *stream) {
    if (n_problems == 0) return DFINE_OK;
    if (!table || n_problems < 0 || max_blocks < 1) return DFINE_E_BADARG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad1_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kW2Ring * 2 * 128 * 128);
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_wgrad1_group_kernel, dim3(max_blocks, n_problems), dim3(kW2Threads), (size_t)kW2Ring * 2 * 128 * 128,
                       (hipStream_t)stream, (const int64_t *)table);
    return check_launch();
}

// table: device int64 [n_problems][8] (rows from dfine_linear_wgrad_group_row); max_blocks = the largest row's workgroup count.
int dfine_linear_wgrad_group(const void *table, int n_problems, int max_blocks, void *stream) {
    if (n_problems == 0) return DFINE_OK;
    if (!table || n_problems < 0 || max_blocks < 1) return DFINE_E_BADARG;
    hipLaunchKernelGGL(linear_wgrad_group_kernel, dim3(max_blocks, n_problems), dim3(kConvThreads), 0, (hipStream_t)stream,
                       (const int64_t *)table);
    return check_launch();
}

// table: device int64 [n_entries][8] = {partials, dst (f32, ACCUMULATED into), splits, Cout, Cin, taps, NP16, CP16};
// max_blocks = the largest dfine_multi_wgrad_reduce_blocks(splits, Cout * Cin * taps) of the rows
int dfine_multi_wgrad_reduce_blocks(int splits, int64_t elems) {
    const int64_t per = (int64_t)wr_groups(splits) * 64;
    return (int)((elems + per - 1) / per);
}

int dfine_multi_wgrad_reduce(const void *table, int n_entries, int max_blocks, void *stream) {
    if (n_entries == 0) return DFINE_OK;
    if (!table || n_entries < 0 || max_blocks < 1) return DFINE_E_BADARG;
    hipLaunchKernelGGL(multi_wgrad_reduce_kernel, dim3(max_blocks, n_entries), dim3(256), 0, (hipStream_t)stream, (const int64_t *)table);
    return check_launch();
}

}  // extern "C"
