// A1 - HGNetv2 stem (src/d_fine/arch/hgnetv2.py:115-166): the five convolutions in front of stage 1 have
// 3..48 channels on 640x640..160x160 planes.  They are pure bandwidth layers (236-354 MB of bf16
// activations, 2-17 GFLOP), but MIOpen serves them with generic kernels at 15-30 TFLOP/s: 1.65 ms forward
// and 4.2 ms backward per D-FINE-m step (profiles/r01_conv_survey_hip_vs_miopen.txt), plus two F.pad
// copies and a max-pool over the padded map.  Here:
//   * stem_conv_kernel      direct convolution, one thread per output pixel holding all output channels in
//                           registers; weights are wave-uniform and come through the scalar cache.  Reads
//                           outside the plane return 0, which also implements F.pad(x, (0,1,0,1)) in front
//                           of the 2x2 convs.  With flipped/transposed weights the same kernel is the data
//                           gradient of the stride-1 layers.
//   * stem_dgrad_s2_kernel  data gradient of the 3x3 stride-2 layer: a thread owns two adjacent input
//                           pixels of one row, so the tap set is uniform per block (row parity) and fixed
//                           per pixel of the pair.
//   * stem_wgrad_kernel     weight gradient on the MFMA units: M = output channels, N = (input channel,
//                           tap) columns, K = 32 consecutive output pixels; every wave is an independent
//                           worker (column group, pixel range) and writes fp32 partial sums.
//   * stem_pool_*           2x2 stride-1 max-pool over the zero-padded map, forward and backward
//                           (argmax recomputed from the input, first maximum in scan order like ATen).
#include "common.h"

namespace dfine {

constexpr int kStemThreads = 256;

// stem3.hip: the 3x3 / stride-2 layer over a two-tensor input on the matrix cores (DFINE_E_BADARG = shape not served)
int stem3_fwd_rows(const uint16_t *xa, const uint16_t *xb, int Ca, const float *wp, uint16_t *y, int B, int Cin, int Cout, int H, int W,
                   int Ho, int Wo, hipStream_t st);
int stem3_bwd_rows(const uint16_t *dy, const float *wq, uint16_t *dxa, uint16_t *dxb, int Ca, int B, int Cin, int Cout, int Ho, int Wo,
                   hipStream_t st);
typedef __attribute__((ext_vector_type(8))) __bf16 stem_bf16x8;
typedef __attribute__((ext_vector_type(4))) float stem_f32x4;

// mode 0: wp[(ci*KS+ky)*KS+kx][co] = w[co][ci][ky][kx]                        (forward)
// mode 1: wp[(co*KS+ky)*KS+kx][ci] = w[co][ci][KS-1-ky][KS-1-kx]              (stride-1 data gradient)
// mode 2: wp[(co*KS+ky)*KS+kx][ci] = w[co][ci][ky][kx]                        (stride-2 data gradient)
__global__ void stem_pack_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int KS,
                                 int mode) {
    const int total = Cout * Cin * KS * KS;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kx = i % KS, ky = (i / KS) % KS, ci = (i / (KS * KS)) % Cin, co = i / (KS * KS * Cin);
    const float v = w[i];
    if (mode == 0) wp[((ci * KS + ky) * KS + kx) * Cout + co] = v;
    else if (mode == 1) wp[((co * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)) * Cin + ci] = v;
    else wp[((co * KS + ky) * KS + kx) * Cin + ci] = v;
}

// y[b][co][yo][xo] = sum_{ci,ky,kx} x[b][ci][yo*S+ky-pad][xo*S+kx-pad] * wp[(ci,ky,kx)][co]   (0 outside the plane)
template <int CIN, int COUT, int KS, int S>
__global__ __launch_bounds__(kStemThreads) void stem_conv_kernel(const uint16_t *__restrict__ x,
                                                                 const uint16_t *__restrict__ x2, int CA,
                                                                 const float *__restrict__ wp,
                                                                 uint16_t *__restrict__ y, int H, int W, int Ho,
                                                                 int Wo, int pad) {
    // x2 != nullptr: the input is the channel concatenation [x (CA channels) | x2 (CIN - CA)] read in place (StemBlock's
    // torch.cat of the pooled stem1 map and the stem2 branch, ref hgnetv2.py:158-163)
    const int p = blockIdx.x * kStemThreads + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    const int64_t HWi = (int64_t)H * W;
    const uint16_t *xa_b = x + (int64_t)b * (x2 ? CA : CIN) * HWi;
    const uint16_t *xb_b = x2 ? x2 + (int64_t)b * (CIN - CA) * HWi : nullptr;
    auto chan = [&](int ci) -> const uint16_t * { return (x2 && ci >= CA) ? xb_b + (int64_t)(ci - CA) * HWi : xa_b + (int64_t)ci * HWi; };
    const uint16_t *xb = xa_b;
    const int yi0 = yo * S - pad, xi0 = xo * S - pad;
    // tap offsets / validity do not depend on the channel; loads are unconditional (clamped address, value
    // zeroed afterwards) so that the KS*KS loads of a channel are all in flight before the first FMA
    int toff[KS * KS];
    bool tok[KS * KS];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int yi = yi0 + ky, xi = xi0 + kx;
            const bool ok = (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W;
            tok[ky * KS + kx] = ok;
            toff[ky * KS + kx] = ok ? yi * W + xi : 0;
        }
    }
    uint16_t raw[KS * KS], nxt[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) raw[t] = xb[toff[t]];
    for (int ci = 0; ci < CIN; ++ci) {
        const float *wc = wp + ci * (KS * KS * COUT);           // wave-uniform -> scalar loads
        const uint16_t *xn = chan(min(ci + 1, CIN - 1));                        // next channel's taps in flight
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) nxt[t] = xn[toff[t]];
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
            const float v = tok[t] ? bf16_to_f32(raw[t]) : 0.f;
            const float *wk = wc + t * COUT;
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = fmaf(v, wk[co], acc[co]);
        }
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) raw[t] = nxt[t];
    }
    uint16_t *yb = y + (int64_t)b * COUT * Ho * Wo + p;
#pragma unroll
    for (int co = 0; co < COUT; ++co) yb[(int64_t)co * Ho * Wo] = f32_to_bf16(acc[co]);
}

// 3x3 / stride 2 / pad 1 layers (stem1, stem3) with 16-byte loads and packed FMAs: a thread owns FOUR consecutive output pixels
// of a row = 9 input columns per kernel row: one aligned 16-byte load (columns 2 xo .. 2 xo + 7) plus the element left of
// them, instead of nine 2-byte loads per pixel, and its accumulators are float pairs (v_pk_fma_f32, weight broadcast from a
// scalar).  stem_conv_kernel is bound by unpacked FMAs and 2-byte loads: 247 us for the 17 GFLOP of the 48 -> 24 layer.
typedef float stem_f32x2 __attribute__((ext_vector_type(2)));
template <int CIN, int COUT>
__global__ __launch_bounds__(kStemThreads) void stem_conv_s2_vec_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ x2,
                                                                        int CA, const float *__restrict__ wp, uint16_t *__restrict__ y,
                                                                        int H, int W, int Ho, int Wo) {
    const int q = blockIdx.x * kStemThreads + threadIdx.x;         // quad of output pixels
    const int b = blockIdx.y;
    const int wq = Wo >> 2;
    if (q >= Ho * wq) return;
    const int yo = q / wq, xo = (q - yo * wq) * 4;
    stem_f32x2 acc[COUT][2];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { acc[co][0] = stem_f32x2{0.f, 0.f}; acc[co][1] = stem_f32x2{0.f, 0.f}; }
    const int64_t HWi = (int64_t)H * W;
    const uint16_t *xa_b = x + (int64_t)b * (x2 ? CA : CIN) * HWi;
    const uint16_t *xb_b = x2 ? x2 + (int64_t)b * (CIN - CA) * HWi : nullptr;
    auto chan = [&](int ci) -> const uint16_t * { return (x2 && ci >= CA) ? xb_b + (int64_t)(ci - CA) * HWi : xa_b + (int64_t)ci * HWi; };
    int roff[3];
    bool rok[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = 2 * yo + ky - 1;
        rok[ky] = (unsigned)yi < (unsigned)H;
        roff[ky] = (rok[ky] ? yi : 0) * W + 2 * xo;              // 2 xo is a multiple of 8: 16-byte aligned (W % 8 == 0)
    }
    const bool left = xo > 0;
    uint4 cur[3], nxt[3];
    uint16_t curl[3], nxtl[3];
    {
        const uint16_t *c0 = chan(0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { cur[ky] = *reinterpret_cast<const uint4 *>(c0 + roff[ky]); curl[ky] = left ? c0[roff[ky] - 1] : (uint16_t)0; }
    }
    for (int ci = 0; ci < CIN; ++ci) {
        const float *wc = wp + ci * (9 * COUT);                  // wave-uniform -> scalar loads
        const uint16_t *cn = chan(min(ci + 1, CIN - 1));         // next channel's rows in flight
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { nxt[ky] = *reinterpret_cast<const uint4 *>(cn + roff[ky]); nxtl[ky] = left ? cn[roff[ky] - 1] : (uint16_t)0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float v[9];                                          // columns 2 xo - 1 .. 2 xo + 7
            const uint4 r = cur[ky];
            v[0] = bf16_to_f32(curl[ky]);
            v[1] = __uint_as_float(r.x << 16); v[2] = __uint_as_float(r.x & 0xffff0000u);
            v[3] = __uint_as_float(r.y << 16); v[4] = __uint_as_float(r.y & 0xffff0000u);
            v[5] = __uint_as_float(r.z << 16); v[6] = __uint_as_float(r.z & 0xffff0000u);
            v[7] = __uint_as_float(r.w << 16); v[8] = __uint_as_float(r.w & 0xffff0000u);
            if (!rok[ky]) {
#pragma unroll
                for (int j = 0; j < 9; ++j) v[j] = 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const stem_f32x2 p0 = {v[kx], v[2 + kx]}, p1 = {v[4 + kx], v[6 + kx]};      // output pixels (0, 1) and (2, 3)
                const float *wk = wc + (ky * 3 + kx) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const stem_f32x2 w2 = {wk[co], wk[co]};
                    acc[co][0] = __builtin_elementwise_fma(p0, w2, acc[co][0]);
                    acc[co][1] = __builtin_elementwise_fma(p1, w2, acc[co][1]);
                }
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { cur[ky] = nxt[ky]; curl[ky] = nxtl[ky]; }
    }
    uint16_t *yb = y + (int64_t)b * COUT * Ho * Wo + (int64_t)yo * Wo + xo;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
        *reinterpret_cast<uint2 *>(yb + (int64_t)co * Ho * Wo) = make_uint2(pack_bf16x2(acc[co][0].x, acc[co][0].y), pack_bf16x2(acc[co][1].x, acc[co][1].y));
}

// ---------------------------------------------------------------------------------------------
// The stride-1 stem layers (2x2: stem2a / stem2b and their data gradients; 1x1: stem4) on the matrix cores.
// stem_conv_kernel is bound by unpacked fp32 FMAs and 2-byte loads (the 24 -> 12 layer: 7.5 GFLOP and 235 MB in 141 us = 53
// TFLOP/s, 1.7 TB/s).  The MFMA operands want 8 consecutive K values per lane, and K = (tap, input channel) is strided by a whole
// plane in NCHW - so the input patch of a tile goes through LDS CHANNELS-LAST: 16-byte global loads along a row (8 pixels of one
// channel), eight 2-byte LDS writes to patch[row][col][channel], and then one aligned ds_read_b128 gives a lane the 8 channels of
// its pixel and tap.  GEMM view per tile: M = output pixels (16 per MFMA), N = output channels (16 / 32), K = taps x channel
// groups of 8 (v_mfma_f32_16x16x32_bf16: A lane (pixel l % 16, K chunk l / 16), B lane (channel l % 16, K chunk l / 16), D lane
// holds pixels 4 (l / 16) + i of channel l % 16: four consecutive pixels = one 8-byte store).
// Workgroups are persistent: the next tile's global loads are issued into registers before the MFMAs of the current one
// and written to LDS after them - loads stay in flight through the compute phase instead of load - wait - compute per tile.
// Weights: the fp32 [(ci, ky, kx)][co] array of dfine_stem_pack_weights (modes 0 / 1) is rounded to bf16 B fragments once per
// workgroup (like every other convolution under bf16 autocast: bf16 weights, fp32 accumulation).
typedef uint32_t stem_u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
constexpr int kSmRows = 8, kSmCols = 64;

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load (vmcnt(0)), which
// would end the next tile's prefetch at each barrier
__device__ __forceinline__ void stem_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CIN, int COUT, int KS>
__global__ __launch_bounds__(kStemThreads) void stem_mfma_s1_kernel(const uint16_t *__restrict__ x, const float *__restrict__ wp,
                                                                    uint16_t *__restrict__ y, int H, int W, int Ho, int Wo, int pad,
                                                                    int tiles_x, int tiles_y, int total_tiles) {
    constexpr int CP = (CIN + 7) / 8 * 8, NCG = CP / 8, NCHUNK = KS * KS * NCG, KSTEPS = (NCHUNK + 3) / 4, NT = (COUT + 15) / 16;
    constexpr int PR = kSmRows + KS - 1, PC = kSmCols + KS - 1;
    // 16 bytes of padding after every 8 pixels of a patch row: the 8-pixel vectors of neighbouring lanes then start 25 (not 24) 16-byte
    // slots apart and the eight ds_write_b128 of a staging item spread over all banks (24 = 8 mod 16 put 8 lanes on 2 slots)
    constexpr int RP = PC * CP + (PC + 7) / 8 * 8;
    // staging item = (8 pixels of a row) x (8 channels): 8 coalesced 16-byte loads (one per channel), an 8 x 8 transpose of bf16
    // in registers (32 v_perm_b32), 8 ds_write_b128 (one pixel's 8 channels each).  [Writing the patch element by element
    // - 56 ds_write_b16 per thread and tile - made the kernel LDS-bound: SQ_LDS_BANK_CONFLICT was 84 % of the LDS-active cycles.]
    constexpr int NV = PR * NCG * (kSmCols / 8), NPF = (NV + kStemThreads - 1) / kStemThreads;
    constexpr int NH = PR * CIN * (KS - 1), NPH = (NH + kStemThreads - 1) / kStemThreads;
    extern __shared__ __attribute__((aligned(16))) uint16_t patch[];          // [PR][PC][CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (CP != CIN && KS > 1) {                                                 // halo column: its pad channels are never written - no NaN bits there
        for (int i = tid; i < PR * RP; i += kStemThreads) patch[i] = 0;
        __syncthreads();
    }
    // B fragments: channel co = nt * 16 + lane % 16, K chunk q = 4 ks + lane / 16 -> (tap, channel group)
    uint4 wreg[KSTEPS][NT];
    int koff[KSTEPS];
    bool kval[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int q = 4 * ks + (lane >> 4);
        const bool qv = q < NCHUNK;
        const int tap = qv ? q / NCG : 0, cig = qv ? q - tap * NCG : 0;
        const int ky = tap / KS, kx = tap - ky * KS;
        kval[ks] = qv;
        const int qx = (lane & 15) + kx;                            // pixel of the lane inside its 16-pixel group, plus the tap's column
        koff[ks] = ky * RP + qx * CP + (qx >> 3) * 8 + cig * 8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + (lane & 15);
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = cig * 8 + j;
                wv[j] = (qv && co < COUT && ci < CIN) ? wp[((ci * KS + ky) * KS + kx) * COUT + co] : 0.f;
            }
            wreg[ks][nt] = make_uint4(pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3]), pack_bf16x2(wv[4], wv[5]),
                                      pack_bf16x2(wv[6], wv[7]));
        }
    }
    uint4 pf[NPF][8];
    uint16_t ph[NPH > 0 ? NPH : 1];
    // what a thread stages is the same for every tile: offsets relative to the tile origin, computed once
    int rel[NPF], lo[NPF], rr[NPF], vv[NPF], cg[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int i = tid + j * kStemThreads;
        const int v = i % (kSmCols / 8), rc = i / (kSmCols / 8), r = rc % PR, g = (i < NV) ? rc / PR : 0;
        rel[j] = (g * 8 * H + r) * W + 8 * v;                       // channel g * 8 (+ c * H * W for the c-th of the group)
        lo[j] = r * RP + 8 * v * CP + v * 8 + g * 8;
        rr[j] = (i < NV) ? r : -(1 << 20);                          // (an item past the patch fails every row test)
        vv[j] = 8 * v;
        cg[j] = g * 8;
    }
    int hrel[NPH > 0 ? NPH : 1], hlo[NPH > 0 ? NPH : 1], hrr[NPH > 0 ? NPH : 1], hk[NPH > 0 ? NPH : 1];
#pragma unroll
    for (int j = 0; j < NPH; ++j) {
        const int i = tid + j * kStemThreads;
        const int k = i % (KS - 1 > 0 ? KS - 1 : 1), rc = i / (KS - 1 > 0 ? KS - 1 : 1), r = rc % PR, ci = (i < NH) ? rc / PR : 0;
        hrel[j] = (ci * H + r) * W + kSmCols + k;
        hlo[j] = r * RP + (kSmCols + k) * CP + ((kSmCols + k) >> 3) * 8 + ci;
        hrr[j] = (i < NH) ? r : -(1 << 20);
        hk[j] = kSmCols + k;
    }
    const int HW = H * W;
    auto coords = [&](int t, int &b, int &y0, int &x0) {
        b = t / (tiles_y * tiles_x);
        const int r = t - b * (tiles_y * tiles_x);
        const int ty = r / tiles_x;
        y0 = ty * kSmRows; x0 = (r - ty * tiles_x) * kSmCols;
    };
    auto fetch = [&](int t) {
        int b, y0, x0;
        coords(t, b, y0, x0);
        const int gy0 = y0 - pad, gx0 = x0 - pad;
        const uint16_t *base = x + (int64_t)b * CIN * H * W + (int64_t)gy0 * W + gx0;     // dereferenced at valid positions only
        const bool cols_in = gx0 >= 0 && gx0 + kSmCols <= W;       // uniform: every 8-pixel vector of the tile lies inside the rows
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const bool row_ok = (unsigned)(gy0 + rr[j]) < (unsigned)H;
            const uint16_t *src = base + rel[j];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pf[j][c] = make_uint4(0, 0, 0, 0);
                if (row_ok && (CP == CIN || cg[j] + c < CIN)) {
                    if (cols_in) {
                        const stem_u32x4_u t4 = *reinterpret_cast<const stem_u32x4_u *>(src + c * HW);
                        pf[j][c] = make_uint4(t4.x, t4.y, t4.z, t4.w);
                    } else {
                        const int gx = gx0 + vv[j];
                        uint16_t e[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) e[k] = (gx + k >= 0 && gx + k < W) ? src[c * HW + k] : (uint16_t)0;
                        pf[j][c] = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16),
                                              e[6] | ((uint32_t)e[7] << 16));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NPH; ++j) {
            ph[j] = 0;
            if ((unsigned)(gy0 + hrr[j]) < (unsigned)H && (unsigned)(gx0 + hk[j]) < (unsigned)W) ph[j] = base[hrel[j]];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            if (rr[j] >= 0) {
                uint16_t *d = patch + lo[j];
                const uint32_t(*w)[4] = reinterpret_cast<const uint32_t(*)[4]>(&pf[j][0]);      // w[channel][pixel pair]
#pragma unroll
                for (int pix = 0; pix < 8; ++pix) {
                    const uint32_t sel = (pix & 1) ? 0x07060302u : 0x05040100u;                  // (b.half << 16) | a.half of perm(b, a)
                    const int dw = pix >> 1;
                    uint4 o;
                    o.x = __builtin_amdgcn_perm(w[1][dw], w[0][dw], sel);
                    o.y = __builtin_amdgcn_perm(w[3][dw], w[2][dw], sel);
                    o.z = __builtin_amdgcn_perm(w[5][dw], w[4][dw], sel);
                    o.w = __builtin_amdgcn_perm(w[7][dw], w[6][dw], sel);
                    *reinterpret_cast<uint4 *>(d + pix * CP) = o;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NPH; ++j)
            if (hrr[j] >= 0) patch[hlo[j]] = ph[j];
    };
    // per-lane parts of the MFMA phase: A fragment row of the lane inside a 16-pixel group, output offset of its 4 pixels
    int orel[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) orel[nt] = (nt * 16 + (lane & 15)) * Ho * Wo + 4 * (lane >> 4);
    const bool wo4 = (Wo & 3) == 0;
    int t = blockIdx.x;
    if (t < total_tiles) fetch(t);
    for (; t < total_tiles; t += gridDim.x) {
        commit();
        stem_lds_barrier();
        const int tn = t + gridDim.x;
        if (tn < total_tiles) fetch(tn);                           // in flight during the MFMAs below
        int b, y0, x0;
        coords(t, b, y0, x0);
        uint16_t *yb = y + (int64_t)b * COUT * Ho * Wo;
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int g = wave * 8 + gi;
            const int tr = g / (kSmCols / 16), c16 = g % (kSmCols / 16);      // wave-uniform
            stem_f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = stem_f32x4{0.f, 0.f, 0.f, 0.f};
            const uint16_t *pp = patch + tr * RP + c16 * (16 * CP + 16);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                uint4 av = make_uint4(0, 0, 0, 0);
                if (kval[ks]) av = *reinterpret_cast<const uint4 *>(pp + koff[ks]);
                const stem_bf16x8 a = __builtin_bit_cast(stem_bf16x8, av);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(stem_bf16x8, wreg[ks][nt]), acc[nt], 0, 0, 0);
            }
            const int gy = y0 + tr, gxb = x0 + c16 * 16;           // uniform
            if (gy < Ho) {
                uint16_t *og = yb + (int64_t)gy * Wo + gxb;
                const int gx = gxb + 4 * (lane >> 4);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt * 16 + (lane & 15) >= COUT) continue;
                    uint16_t *o = og + orel[nt];
                    if (wo4 && gx + 3 < Wo) {
                        *reinterpret_cast<uint2 *>(o) = make_uint2(pack_bf16x2(acc[nt][0], acc[nt][1]), pack_bf16x2(acc[nt][2], acc[nt][3]));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (gx + i < Wo) o[i] = f32_to_bf16(acc[nt][i]);
                    }
                }
            }
        }
        stem_lds_barrier();                                         // every wave is done with the patch before it is overwritten
    }
}

// dx[b][ci][yi][xi] of a 3x3 / stride 2 / pad 1 convolution; dy [B, COUT, Ho, Wo], H = 2*Ho, W = 2*Wo.
// wq[((co*3+ky)*3+kx)][ci].  thread = (row yi, pixel pair 2c / 2c+1).
template <int CIN, int COUT>
__global__ __launch_bounds__(kStemThreads) void stem_dgrad_s2_kernel(const uint16_t *__restrict__ dy,
                                                                     const float *__restrict__ wq,
                                                                     uint16_t *__restrict__ dx, uint16_t *__restrict__ dx2,
                                                                     int CA, int H, int W, int Ho, int Wo) {
    // dx2 != nullptr: channels [0, CA) of the gradient go to dx, [CA, CIN) to dx2 (one contiguous tensor per concatenated input)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int yi = blockIdx.y, b = blockIdx.z;
    if (c >= Wo) return;
    float a0[CIN], a1[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) { a0[ci] = 0.f; a1[ci] = 0.f; }
    const bool odd = yi & 1;
    const int ntap = odd ? 2 : 1;
    const uint16_t *dyb = dy + (int64_t)b * COUT * Ho * Wo;
    const bool has1 = c + 1 < Wo;
    const int c1 = has1 ? c + 1 : c;
    // rows of dy feeding this input row: (yi + 1 - ky) even -> ky = 1 (even yi) or ky in {0, 2} (odd yi)
    int ky_t[2], yo_t[2];
    bool ok_t[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ky_t[t] = odd ? 2 * t : 1;
        yo_t[t] = (yi + 1 - ky_t[t]) >> 1;
        ok_t[t] = t < ntap && yo_t[t] >= 0 && yo_t[t] < Ho;      // uniform per block
        yo_t[t] = min(max(yo_t[t], 0), Ho - 1);
    }
    const int total = COUT * ntap;
    uint16_t r0 = dyb[(int64_t)yo_t[0] * Wo + c], r1 = dyb[(int64_t)yo_t[0] * Wo + c1];
    for (int it = 0; it < total; ++it) {
        const int co = odd ? it >> 1 : it, t = odd ? it & 1 : 0;
        // next (co, tap) pair's two loads are issued before this pair's FMAs
        const int itn = min(it + 1, total - 1);
        const int con = odd ? itn >> 1 : itn, tn = odd ? itn & 1 : 0;
        const uint16_t *rown = dyb + ((int64_t)con * Ho + yo_t[tn]) * Wo;
        const uint16_t n0 = rown[c], n1 = rown[c1];
        if (ok_t[t]) {
            const float d0 = bf16_to_f32(r0);
            const float d1 = has1 ? bf16_to_f32(r1) : 0.f;
            const float *wk = wq + (co * 3 + ky_t[t]) * 3 * CIN;  // wave-uniform
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                a0[ci] = fmaf(d0, wk[CIN + ci], a0[ci]);                           // xi = 2c   : kx = 1, xo = c
                a1[ci] = fmaf(d1, wk[ci], fmaf(d0, wk[2 * CIN + ci], a1[ci]));     // xi = 2c+1 : kx = 0 (xo = c+1), kx = 2 (xo = c)
            }
        }
        r0 = n0; r1 = n1;
    }
    const int64_t HWi = (int64_t)H * W, rowoff = (int64_t)yi * W + 2 * c;
    uint16_t *oa = dx + (int64_t)b * (dx2 ? CA : CIN) * HWi + rowoff;
    uint16_t *ob = dx2 ? dx2 + (int64_t)b * (CIN - CA) * HWi + rowoff : nullptr;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        uint16_t *o = (dx2 && ci >= CA) ? ob + (int64_t)(ci - CA) * HWi : oa + (int64_t)ci * HWi;
        *reinterpret_cast<uint32_t *>(o) = pack_bf16x2(a0[ci], a1[ci]);
    }
}

// 8 bf16 elements x[xi0 + j*S], j = 0..7, of one plane row as a packed MFMA fragment (0 outside [0, W)).
// One (S = 1) or two (S = 2) 16-byte loads from a start clamped into the row - 2-byte aligned, the hardware
// runs in unaligned-access mode - then a shift by the clamp distance (|d| <= 1 element: pad <= 1, KS <= 3).
template <int S>
__device__ __forceinline__ uint4 stem_row8(const uint16_t *rowp, int xi0, int W, bool row_ok) {
    constexpr int SPAN = 8 * S;
    const int xs = min(max(xi0, 0), W - SPAN);
    const int d = xs - xi0;                                      // +1: window starts one element late, -1: one early
    uint4 r;
    if (S == 1) {
        const stem_u32x4_u v = *reinterpret_cast<const stem_u32x4_u *>(rowp + xs);
        if (d == 0) { r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; }
        else if (d > 0) { r.x = v.x << 16; r.y = (v.y << 16) | (v.x >> 16); r.z = (v.z << 16) | (v.y >> 16); r.w = (v.w << 16) | (v.z >> 16); }
        else { r.x = (v.x >> 16) | (v.y << 16); r.y = (v.y >> 16) | (v.z << 16); r.z = (v.z >> 16) | (v.w << 16); r.w = v.w >> 16; }
    } else {
        const stem_u32x4_u a = *reinterpret_cast<const stem_u32x4_u *>(rowp + xs);
        const stem_u32x4_u b = *reinterpret_cast<const stem_u32x4_u *>(rowp + xs + 8);
        if (d == 0) {          // element j = low half of dword j
            r.x = (a.x & 0xffffu) | (a.y << 16); r.y = (a.z & 0xffffu) | (a.w << 16);
            r.z = (b.x & 0xffffu) | (b.y << 16); r.w = (b.z & 0xffffu) | (b.w << 16);
        } else if (d < 0) {    // element j = high half of dword j
            r.x = (a.x >> 16) | (a.y & 0xffff0000u); r.y = (a.z >> 16) | (a.w & 0xffff0000u);
            r.z = (b.x >> 16) | (b.y & 0xffff0000u); r.w = (b.z >> 16) | (b.w & 0xffff0000u);
        } else {               // element 0 is outside, element j = high half of dword j - 1
            r.x = a.x & 0xffff0000u;                 r.y = (a.y >> 16) | (a.z & 0xffff0000u);
            r.z = (a.w >> 16) | (b.x & 0xffff0000u); r.w = (b.y >> 16) | (b.z & 0xffff0000u);
        }
    }
    if (!row_ok) r = make_uint4(0, 0, 0, 0);
    return r;
}

// dw[co][ci][ky][kx] = sum_{b,yo,xo} dy[b][co][yo][xo] * x[b][ci][yo*S+ky-pad][xo*S+kx-pad].
// part[split][co][n], n = (ci*KS+ky)*KS+kx.  Wo % 32 == 0, COUT <= 32.
constexpr int kStemNT = 4;          // 16-column tiles per worker
template <int S>
__global__ __launch_bounds__(kStemThreads) void stem_wgrad_kernel(const uint16_t *__restrict__ x,
                                                                  const uint16_t *__restrict__ x2, int CA,
                                                                  const uint16_t *__restrict__ dy,
                                                                  float *__restrict__ part, int CIN, int COUT, int KS,
                                                                  int pad, int H, int W, int Ho, int Wo,
                                                                  int ngroups, int nsplit, int total_steps,
                                                                  int steps_per_split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int worker = blockIdx.x * (kStemThreads / 64) + wave;
    const int g = worker % ngroups, split = worker / ngroups;
    if (split >= nsplit) return;
    const int ncols = CIN * KS * KS;
    const int kk = KS * KS;
    int col_off[kStemNT], col_ky[kStemNT], col_kx[kStemNT];
    bool col_b[kStemNT];                                         // the column's channel lives in the second source
#pragma unroll
    for (int t = 0; t < kStemNT; ++t) {
        const int n = (g * kStemNT + t) * 16 + (lane & 15);
        const bool ok = n < ncols;
        const int ci = ok ? n / kk : 0, tap = ok ? n - ci * kk : 0;
        col_b[t] = x2 != nullptr && ci >= CA;
        col_off[t] = ok ? (col_b[t] ? ci - CA : ci) * H * W : -1;
        col_ky[t] = tap / KS;
        col_kx[t] = tap - (tap / KS) * KS;
    }
    stem_f32x4 acc[kStemNT][2];
#pragma unroll
    for (int t = 0; t < kStemNT; ++t) { acc[t][0] = stem_f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = stem_f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int wsteps = Wo / 32;
    const int kg = lane >> 4;
    const int s0 = split * steps_per_split, s1 = min(total_steps, s0 + steps_per_split);
    // Software pipeline: the loads of step s + 1 (two dy fragments, one or two 16-byte pieces of x per column tile) are issued
    // before the MFMAs of step s.  [Issued and consumed in the same step, every wave sat out a full memory latency per 8 MFMAs:
    // SQ_WAIT_ANY was 74 % of the wave cycles of the 48 -> 24 layer, 360 us for 17 GFLOP and 354 MB.]  The step's position
    // (image, row, 32-pixel column block) is advanced, not divided.
    // x fragments: ALIGNED 16-byte loads of the 8 (stride 1) / 16 (stride 2) input pixels under the lane's 8 output pixels plus, for
    // a tap that sits one column left / right of them, one 2-byte load of the neighbouring element; the fragment is assembled
    // with shifts.  (Loading the shifted window directly - 16 bytes at a 2-byte-aligned address - works in the hardware's
    // unaligned mode but at a fraction of the load rate: the 48 -> 24 layer issues 1.8 GB of such loads.)
    struct Stage {
        uint4 a[2];
        uint4 raw[kStemNT][S];
        uint16_t e[kStemNT];
        bool ok[kStemNT];
    };
    int coff[kStemNT];                                           // tap column relative to the aligned window: -1, 0, +1
#pragma unroll
    for (int t = 0; t < kStemNT; ++t) coff[t] = col_kx[t] - pad;
    int pb = s0 / (Ho * wsteps), pr = s0 - pb * (Ho * wsteps), pyo = pr / wsteps, pxw = (pr - pyo * wsteps) * 32;
    auto issue = [&](Stage &st) {
        const int xo0 = pxw + 8 * kg;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = mt * 16 + (lane & 15);
            st.a[mt] = make_uint4(0, 0, 0, 0);
            if (co < COUT) st.a[mt] = *reinterpret_cast<const uint4 *>(dy + (((int64_t)pb * COUT + co) * Ho + pyo) * Wo + xo0);
        }
        const uint16_t *xb_a = x + (int64_t)pb * (x2 ? CA : CIN) * H * W;
        const uint16_t *xb_b = x2 ? x2 + (int64_t)pb * (CIN - CA) * H * W : xb_a;
        // multiple of 8 elements: 16-byte aligned when W % 8 == 0 (any other width still works, at the unaligned rate); past the row only
        // under zero-padded dy columns (hip._stem_pad32), where the values do not matter: clamped into the row
        const int base = min(xo0 * S, W - 8 * S);
#pragma unroll
        for (int t = 0; t < kStemNT; ++t) {
            if ((g * kStemNT + t) * 16 >= ncols) break;          // uniform
            const int yi = pyo * S + col_ky[t] - pad;
            const bool row_ok = col_off[t] >= 0 && (unsigned)yi < (unsigned)H;
            const uint16_t *rowp = (col_b[t] ? xb_b : xb_a) + (row_ok ? col_off[t] + yi * W : 0);
            st.ok[t] = row_ok;
            const stem_u32x4_u r0 = *reinterpret_cast<const stem_u32x4_u *>(rowp + base);
            st.raw[t][0] = make_uint4(r0.x, r0.y, r0.z, r0.w);
            if (S == 2) {
                const stem_u32x4_u r1 = *reinterpret_cast<const stem_u32x4_u *>(rowp + base + 8);
                st.raw[t][S - 1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
            }
            st.e[t] = 0;
            if (coff[t] < 0) { if (base > 0) st.e[t] = rowp[base - 1]; }
            else if (S == 1 && coff[t] > 0) { if (base + 8 < W) st.e[t] = rowp[base + 8]; }
        }
        pxw += 32;                                               // next step
        if (pxw >= Wo) { pxw = 0; if (++pyo == Ho) { pyo = 0; ++pb; } }
    };
    auto consume = [&](const Stage &st) {
        const stem_bf16x8 a0 = __builtin_bit_cast(stem_bf16x8, st.a[0]), a1 = __builtin_bit_cast(stem_bf16x8, st.a[1]);
#pragma unroll
        for (int t = 0; t < kStemNT; ++t) {
            if ((g * kStemNT + t) * 16 >= ncols) break;          // uniform
            const int d = coff[t];
            const uint32_t e = st.e[t];
            uint4 r;
            if (S == 1) {
                const uint4 v = st.raw[t][0];
                if (d == 0) r = v;
                else if (d < 0) { r.x = (v.x << 16) | e; r.y = (v.y << 16) | (v.x >> 16); r.z = (v.z << 16) | (v.y >> 16); r.w = (v.w << 16) | (v.z >> 16); }
                else { r.x = (v.x >> 16) | (v.y << 16); r.y = (v.y >> 16) | (v.z << 16); r.z = (v.z >> 16) | (v.w << 16); r.w = (v.w >> 16) | (e << 16); }
            } else {
                const uint4 p = st.raw[t][0], q = st.raw[t][S - 1];
                if (d == 0) {          // elements 0, 2, .. 14: low half of every dword
                    r.x = (p.x & 0xffffu) | (p.y << 16); r.y = (p.z & 0xffffu) | (p.w << 16);
                    r.z = (q.x & 0xffffu) | (q.y << 16); r.w = (q.z & 0xffffu) | (q.w << 16);
                } else if (d > 0) {    // elements 1, 3, .. 15: high half of every dword
                    r.x = (p.x >> 16) | (p.y & 0xffff0000u); r.y = (p.z >> 16) | (p.w & 0xffff0000u);
                    r.z = (q.x >> 16) | (q.y & 0xffff0000u); r.w = (q.z >> 16) | (q.w & 0xffff0000u);
                } else {               // elements -1, 1, .. 13: the element left of the window, then the high halves of dwords 0 .. 6
                    r.x = e | (p.x & 0xffff0000u);           r.y = (p.y >> 16) | (p.z & 0xffff0000u);
                    r.z = (p.w >> 16) | (q.x & 0xffff0000u); r.w = (q.y >> 16) | (q.z & 0xffff0000u);
                }
            }
            if (!st.ok[t]) r = make_uint4(0, 0, 0, 0);
            const stem_bf16x8 bf = __builtin_bit_cast(stem_bf16x8, r);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf, acc[t][0], 0, 0, 0);
            if (COUT > 16) acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc[t][1], 0, 0, 0);
        }
    };
    Stage stA, stB;
    if (s0 < s1) issue(stA);
    for (int step = s0; step < s1; step += 2) {
        if (step + 1 < s1) issue(stB);
        consume(stA);
        if (step + 1 >= s1) break;
        if (step + 2 < s1) issue(stA);
        consume(stB);
    }
#pragma unroll
    for (int t = 0; t < kStemNT; ++t) {
        const int n = (g * kStemNT + t) * 16 + (lane & 15);
        if (n >= ncols) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mt * 16 + 4 * kg + r;
                if (co < COUT) part[((int64_t)split * COUT + co) * ncols + n] = acc[t][mt][r];
            }
        }
    }
}

// The same weight gradient with the columns of a worker ordered (pair group, kx): a lane owns ONE (input channel, kernel row) pair
// per group and forms the fragments of all KS kernel columns from the SAME aligned window of that row (+ one edge element) - the
// kernel above gives every (channel, ky, kx) column a lane of its own and loads the window once per kx: 14 vector loads per 8
// MFMAs, 1.8 GB of L1 traffic for the 354 MB of the 48 -> 24 layer (the kernel ran at a third of its HBM time, TA-bound).
// Here: 2 + PG * (S + 1) loads per 2 * PG * KS MFMAs (11 per 18 for 3x3 / stride 2).  KS in {2, 3}; PAD / S as instantiated.
template <int S, int KS, int PAD, int PG /* pair groups (16 pairs each) per worker */>
__global__ __launch_bounds__(kStemThreads) void stem_wgrad_kx_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ x2, int CA,
                                                                     const uint16_t *__restrict__ dy, float *__restrict__ part, int CIN,
                                                                     int COUT, int H, int W, int Ho, int Wo, int ngroups, int nsplit,
                                                                     int total_steps, int steps_per_split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int worker = blockIdx.x * (kStemThreads / 64) + wave;
    const int g = worker % ngroups, split = worker / ngroups;
    if (split >= nsplit) return;
    const int npairs = CIN * KS, ncols = npairs * KS;
    int pr_off[PG], pr_ky[PG];
    bool pr_b[PG];                                               // the pair's channel lives in the second source
#pragma unroll
    for (int q = 0; q < PG; ++q) {
        const int p = (g * PG + q) * 16 + (lane & 15);
        const bool ok = p < npairs;
        const int ci = ok ? p / KS : 0;
        pr_ky[q] = ok ? p - ci * KS : 0;
        pr_b[q] = x2 != nullptr && ci >= CA;
        pr_off[q] = ok ? (pr_b[q] ? ci - CA : ci) * H * W : -1;
    }
    stem_f32x4 acc[PG][KS][2];
#pragma unroll
    for (int q = 0; q < PG; ++q)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) { acc[q][kx][0] = stem_f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][kx][1] = stem_f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int wsteps = Wo / 32;
    const int kg = lane >> 4;
    const int s0 = split * steps_per_split, s1 = min(total_steps, s0 + steps_per_split);
    struct Stage {
        uint4 a[2];
        uint4 raw[PG][S];
        uint32_t el[PG], er[PG];                                 // the element left of the window / right of it (stride 1)
        bool ok[PG];
    };
    int pb = s0 / (Ho * wsteps), pr = s0 - pb * (Ho * wsteps), pyo = pr / wsteps, pxw = (pr - pyo * wsteps) * 32;
    auto issue = [&](Stage &st) {
        const int xo0 = pxw + 8 * kg;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = mt * 16 + (lane & 15);
            st.a[mt] = make_uint4(0, 0, 0, 0);
            if (co < COUT) st.a[mt] = *reinterpret_cast<const uint4 *>(dy + (((int64_t)pb * COUT + co) * Ho + pyo) * Wo + xo0);
        }
        const uint16_t *xb_a = x + (int64_t)pb * (x2 ? CA : CIN) * H * W;
        const uint16_t *xb_b = x2 ? x2 + (int64_t)pb * (CIN - CA) * H * W : xb_a;
        const int base = min(xo0 * S, W - 8 * S);                // (see stem_wgrad_kernel)
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            if ((g * PG + q) * 16 >= npairs) break;              // uniform
            const int yi = pyo * S + pr_ky[q] - PAD;
            const bool row_ok = pr_off[q] >= 0 && (unsigned)yi < (unsigned)H;
            const uint16_t *rowp = (pr_b[q] ? xb_b : xb_a) + (row_ok ? pr_off[q] + yi * W : 0);
            st.ok[q] = row_ok;
            const stem_u32x4_u r0 = *reinterpret_cast<const stem_u32x4_u *>(rowp + base);
            st.raw[q][0] = make_uint4(r0.x, r0.y, r0.z, r0.w);
            if (S == 2) {
                const stem_u32x4_u r1 = *reinterpret_cast<const stem_u32x4_u *>(rowp + base + 8);
                st.raw[q][S - 1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
            }
            st.el[q] = 0; st.er[q] = 0;
            if (PAD > 0 && base > 0) st.el[q] = rowp[base - 1];
            if (S == 1 && KS - 1 - PAD > 0 && base + 8 < W) st.er[q] = rowp[base + 8];
        }
        pxw += 32;
        if (pxw >= Wo) { pxw = 0; if (++pyo == Ho) { pyo = 0; ++pb; } }
    };
    auto consume = [&](const Stage &st) {
        const stem_bf16x8 a0 = __builtin_bit_cast(stem_bf16x8, st.a[0]), a1 = __builtin_bit_cast(stem_bf16x8, st.a[1]);
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            if ((g * PG + q) * 16 >= npairs) break;              // uniform
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                constexpr int dummy = 0; (void)dummy;
                const int d = kx - PAD;                           // compile-time per unrolled kx
                uint4 r;
                if (S == 1) {
                    const uint4 v = st.raw[q][0];
                    if (d == 0) r = v;
                    else if (d < 0) { r.x = (v.x << 16) | st.el[q]; r.y = (v.y << 16) | (v.x >> 16); r.z = (v.z << 16) | (v.y >> 16); r.w = (v.w << 16) | (v.z >> 16); }
                    else { r.x = (v.x >> 16) | (v.y << 16); r.y = (v.y >> 16) | (v.z << 16); r.z = (v.z >> 16) | (v.w << 16); r.w = (v.w >> 16) | (st.er[q] << 16); }
                } else {
                    const uint4 p = st.raw[q][0], w2 = st.raw[q][S - 1];
                    if (d == 0) {
                        r.x = (p.x & 0xffffu) | (p.y << 16); r.y = (p.z & 0xffffu) | (p.w << 16);
                        r.z = (w2.x & 0xffffu) | (w2.y << 16); r.w = (w2.z & 0xffffu) | (w2.w << 16);
                    } else if (d > 0) {
                        r.x = (p.x >> 16) | (p.y & 0xffff0000u); r.y = (p.z >> 16) | (p.w & 0xffff0000u);
                        r.z = (w2.x >> 16) | (w2.y & 0xffff0000u); r.w = (w2.z >> 16) | (w2.w & 0xffff0000u);
                    } else {
                        r.x = st.el[q] | (p.x & 0xffff0000u);      r.y = (p.y >> 16) | (p.z & 0xffff0000u);
                        r.z = (p.w >> 16) | (w2.x & 0xffff0000u);  r.w = (w2.y >> 16) | (w2.z & 0xffff0000u);
                    }
                }
                if (!st.ok[q]) r = make_uint4(0, 0, 0, 0);
                const stem_bf16x8 bf = __builtin_bit_cast(stem_bf16x8, r);
                acc[q][kx][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf, acc[q][kx][0], 0, 0, 0);
                if (COUT > 16) acc[q][kx][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc[q][kx][1], 0, 0, 0);
            }
        }
    };
    Stage stA, stB;
    if (s0 < s1) issue(stA);
    for (int step = s0; step < s1; step += 2) {
        if (step + 1 < s1) issue(stB);
        consume(stA);
        if (step + 1 >= s1) break;
        if (step + 2 < s1) issue(stA);
        consume(stB);
    }
#pragma unroll
    for (int q = 0; q < PG; ++q) {
        const int p = (g * PG + q) * 16 + (lane & 15);
        if (p >= npairs) continue;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = mt * 16 + 4 * kg + r;
                    if (co < COUT) part[((int64_t)split * COUT + co) * ncols + p * KS + kx] = acc[q][kx][mt][r];
                }
    }
}

// dw (zeroed by the caller) += sum over this block's chunk of splits; 64 columns x 4 split lanes per block
__global__ void stem_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, int nsplit, int total,
                                         int splits_per_block) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const int k0 = blockIdx.y * splits_per_block, k1 = min(nsplit, k0 + splits_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (i < total) {
        int k = k0 + q;
        for (; k + 4 < k1; k += 8) { s0 += part[(int64_t)k * total + i]; s1 += part[(int64_t)(k + 4) * total + i]; }
        for (; k < k1; k += 4) s0 += part[(int64_t)k * total + i];
    }
    red[q][col] = s0 + s1;
    __syncthreads();
    if (q == 0 && i < total) unsafeAtomicAdd(dw + i, (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]));
}

// 2x2 / stride 1 max-pool over the map padded by one zero row / column at the bottom / right.
__global__ void stem_pool_fwd_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int H, int W, int64_t planes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int64_t pl = i / (H * W);
    const int r = (int)(i - pl * H * W);
    const int yy = r / W, xx = r - yy * W;
    const uint16_t *p = x + pl * H * W;
    const bool hx = xx + 1 < W, hy = yy + 1 < H;
    const int y1 = hy ? yy + 1 : yy, x1 = hx ? xx + 1 : xx;
    const uint16_t r00 = p[yy * W + xx], r01 = p[yy * W + x1], r10 = p[y1 * W + xx], r11 = p[y1 * W + x1];
    float m = bf16_to_f32(r00);
    float v = hx ? bf16_to_f32(r01) : 0.f;
    if (v > m || v != v) m = v;
    v = hy ? bf16_to_f32(r10) : 0.f;
    if (v > m || v != v) m = v;
    v = (hx && hy) ? bf16_to_f32(r11) : 0.f;
    if (v > m || v != v) m = v;
    y[i] = f32_to_bf16(m);
}

// position (0..3) of the first maximum among (a, b, c, d) = window scan order, strict >
__device__ __forceinline__ int first_max4(float a, float b, float c, float d) {
    float m = a;
    int k = 0;
    if (b > m || b != b) { m = b; k = 1; }
    if (c > m || c != c) { m = c; k = 2; }
    if (d > m || d != d) { k = 3; }
    return k;
}

__global__ void stem_pool_bwd_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy, uint16_t *__restrict__ dx,
                                     int H, int W, int64_t planes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int64_t pl = i / (H * W);
    const int r = (int)(i - pl * H * W);
    const int yy = r / W, xx = r - yy * W;
    const uint16_t *p = x + pl * H * W, *g = dy + pl * H * W;
    // 3x3 neighbourhood of x around (yy, xx) (0 past the bottom / right edge = the zero pad; rows / columns
    // before the plane only belong to windows that do not exist) and the four window gradients
    const int ym = yy > 0 ? yy - 1 : yy, xm = xx > 0 ? xx - 1 : xx;
    const bool hx = xx + 1 < W, hy = yy + 1 < H;
    const int yp = hy ? yy + 1 : yy, xp = hx ? xx + 1 : xx;
    uint16_t raw[9];
    raw[0] = p[ym * W + xm]; raw[1] = p[ym * W + xx]; raw[2] = p[ym * W + xp];
    raw[3] = p[yy * W + xm]; raw[4] = p[yy * W + xx]; raw[5] = p[yy * W + xp];
    raw[6] = p[yp * W + xm]; raw[7] = p[yp * W + xx]; raw[8] = p[yp * W + xp];
    const uint16_t g00 = g[ym * W + xm], g01 = g[ym * W + xx], g10 = g[yy * W + xm], g11 = g[yy * W + xx];
    float n[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) n[k] = bf16_to_f32(raw[k]);
    if (!hx) { n[2] = 0.f; n[5] = 0.f; n[8] = 0.f; }
    if (!hy) { n[6] = 0.f; n[7] = 0.f; n[8] = 0.f; }
    float s = 0.f;
    // window origin (yy-1, xx-1): (yy, xx) is its position 3; (yy-1, xx): position 2; (yy, xx-1): position 1; (yy, xx): 0
    if (yy > 0 && xx > 0 && first_max4(n[0], n[1], n[3], n[4]) == 3) s += bf16_to_f32(g00);
    if (yy > 0 && first_max4(n[1], n[2], n[4], n[5]) == 2) s += bf16_to_f32(g01);
    if (xx > 0 && first_max4(n[3], n[4], n[6], n[7]) == 1) s += bf16_to_f32(g10);
    if (first_max4(n[4], n[5], n[7], n[8]) == 0) s += bf16_to_f32(g11);
    dx[i] = f32_to_bf16(s);
}

// ---- 8 pixels per thread (W % 8 == 0): one 16-byte load per row instead of eight 2-byte loads -----------------
__device__ __forceinline__ void unpack8(const uint4 &v, float (&o)[8]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&o)[8]) {
    uint4 r;
    r.x = pack_bf16x2(o[0], o[1]);
    r.y = pack_bf16x2(o[2], o[3]);
    r.z = pack_bf16x2(o[4], o[5]);
    r.w = pack_bf16x2(o[6], o[7]);
    return r;
}
// row[-1 .. 8] of a plane row around column x0 (x0 % 8 == 0): e[0] = column x0-1, e[1..8] = x0..x0+7, e[9] = x0+8;
// columns outside [0, W) and rows outside [0, H) read as `fill`
__device__ __forceinline__ void load_row10(const uint16_t *p, int y, int x0, int H, int W, float fill, float (&e)[10]) {
    const bool row_ok = y >= 0 && y < H;
    const int yc = min(max(y, 0), H - 1);
    const uint16_t *r = p + (int64_t)yc * W;
    const uint4 v = *reinterpret_cast<const uint4 *>(r + x0);
    const uint16_t l = r[max(x0 - 1, 0)], rr = r[min(x0 + 8, W - 1)];
    float m[8];
    unpack8(v, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j + 1] = row_ok ? m[j] : fill;
    e[0] = (row_ok && x0 > 0) ? bf16_to_f32(l) : fill;
    e[9] = (row_ok && x0 + 8 < W) ? bf16_to_f32(rr) : fill;
}

__global__ void stem_pool_fwd8_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int H, int W, int64_t planes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // 8-pixel vector index
    const int wv = W >> 3;
    if (i >= planes * H * wv) return;
    const int64_t pl = i / (H * wv);
    const int r = (int)(i - pl * H * wv);
    const int yy = r / wv, x0 = (r - yy * wv) * 8;
    const uint16_t *p = x + pl * H * W;
    float a[10], b[10], o[8];
    load_row10(p, yy, x0, H, W, 0.f, a);
    load_row10(p, yy + 1, x0, H, W, 0.f, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float m = a[j + 1], v = a[j + 2];
        if (v > m || v != v) m = v;
        v = b[j + 1];
        if (v > m || v != v) m = v;
        v = b[j + 2];
        if (v > m || v != v) m = v;
        o[j] = m;
    }
    *reinterpret_cast<uint4 *>(y + pl * H * W + (int64_t)yy * W + x0) = pack8(o);
}

// ACC: dx += the pool's gradient (dx holds the gradient of the map's other consumer: bf16 + bf16 in fp32, one rounding - what
// autograd's add of the two gradient maps would give)
template <bool ACC>
__global__ void stem_pool_bwd8_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy, uint16_t *__restrict__ dx,
                                      int H, int W, int64_t planes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wv = W >> 3;
    if (i >= planes * H * wv) return;
    const int64_t pl = i / (H * wv);
    const int r = (int)(i - pl * H * wv);
    const int yy = r / wv, x0 = (r - yy * wv) * 8;
    const uint16_t *p = x + pl * H * W, *g = dy + pl * H * W;
    float n0[10], n1[10], n2[10], g0[10], g1[10], o[8];
    load_row10(p, yy - 1, x0, H, W, 0.f, n0);       // rows / columns before the plane only feed windows that do not exist
    load_row10(p, yy, x0, H, W, 0.f, n1);
    load_row10(p, yy + 1, x0, H, W, 0.f, n2);
    load_row10(g, yy - 1, x0, H, W, 0.f, g0);
    load_row10(g, yy, x0, H, W, 0.f, g1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = j + 1, xx = x0 + j;             // e[c] = column xx
        float s = 0.f;
        if (yy > 0 && xx > 0 && first_max4(n0[c - 1], n0[c], n1[c - 1], n1[c]) == 3) s += g0[c - 1];
        if (yy > 0 && first_max4(n0[c], n0[c + 1], n1[c], n1[c + 1]) == 2) s += g0[c];
        if (xx > 0 && first_max4(n1[c - 1], n1[c], n2[c - 1], n2[c]) == 1) s += g1[c - 1];
        if (first_max4(n1[c], n1[c + 1], n2[c], n2[c + 1]) == 0) s += g1[c];
        o[j] = s;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(dx + pl * H * W + (int64_t)yy * W + x0);
    if (ACC) {
        const uint4 old = *dst;
        const uint32_t w[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float prev = (j & 1) ? __uint_as_float(w[j >> 1] & 0xffff0000u) : __uint_as_float(w[j >> 1] << 16);
            o[j] = bf16_to_f32(f32_to_bf16(o[j])) + prev;
        }
    }
    *dst = pack8(o);
}

struct StemCfg { int cin, cout, ks, s; };

template <int CIN, int COUT, int KS, int S>
static void launch_stem(const uint16_t *x, const uint16_t *x2, int ca, const float *wp, uint16_t *y, int B, int H, int W,
                        int Ho, int Wo, int pad, hipStream_t st) {
    dim3 grid((Ho * Wo + kStemThreads - 1) / kStemThreads, B);
    hipLaunchKernelGGL((stem_conv_kernel<CIN, COUT, KS, S>), grid, dim3(kStemThreads), 0, st, x, x2, ca, wp, y, H, W, Ho, Wo, pad);
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_stem_supported(int Cin, int Cout, int KS, int stride) {
    // forward configurations (the data gradient of a stride-1 layer is the same kernel with the channel counts swapped)
    static const StemCfg ok[] = {{3, 24, 3, 2}, {24, 12, 2, 1}, {12, 24, 2, 1}, {48, 24, 3, 2}, {24, 32, 1, 1},
                                 {3, 16, 3, 2}, {16, 8, 2, 1}, {8, 16, 2, 1}, {32, 16, 3, 2}, {16, 16, 1, 1},
                                 {3, 32, 3, 2}, {32, 16, 2, 1}, {16, 32, 2, 1}, {64, 32, 3, 2}, {32, 48, 1, 1}, {32, 64, 1, 1},
                                 {32, 24, 1, 1}, {48, 32, 1, 1}, {64, 32, 1, 1}};
    for (const StemCfg &c : ok)
        if (c.cin == Cin && c.cout == Cout && c.ks == KS && c.s == stride) return 1;
    return 0;
}

int dfine_stem_pack_weights(const float *w, float *wp, int Cout, int Cin, int KS, int mode, void *stream) {
    if (!w || !wp || Cout < 1 || Cin < 1 || KS < 1 || mode < 0 || mode > 2) return DFINE_E_BADARG;
    const int total = Cout * Cin * KS * KS;
    hipLaunchKernelGGL(stem_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KS, mode);
    return check_launch();
}

// Direct convolution y[B,Cout,Ho,Wo] = conv(x[B,Cin,H,W]) with zero fill outside the plane; wp from
// dfine_stem_pack_weights(mode 0) - or mode 1 with Cin/Cout exchanged (stride-1 data gradient, pad' = KS-1-pad).
static int stem_conv_impl(const void *x, const void *x2, int ca, const float *wp, void *y, int B, int Cin, int Cout, int H, int W,
                          int Ho, int Wo, int KS, int stride, int pad, void *stream);

int dfine_stem_conv_bf16(const void *x, const float *wp, void *y, int B, int Cin, int Cout, int H, int W, int Ho, int Wo,
                         int KS, int stride, int pad, void *stream) {
    return stem_conv_impl(x, nullptr, Cin, wp, y, B, Cin, Cout, H, W, Ho, Wo, KS, stride, pad, stream);
}

// The same with the input given as two tensors that the reference concatenates along the channels first
// (xa [B, Ca, H, W], xb [B, Cin - Ca, H, W]).
int dfine_stem_conv2_bf16(const void *xa, const void *xb, int Ca, const float *wp, void *y, int B, int Cin, int Cout, int H,
                          int W, int Ho, int Wo, int KS, int stride, int pad, void *stream) {
    if (!xb || Ca < 1 || Ca >= Cin) return DFINE_E_BADARG;
    return stem_conv_impl(xa, xb, Ca, wp, y, B, Cin, Cout, H, W, Ho, Wo, KS, stride, pad, stream);
}

static int stem_conv_impl(const void *x, const void *x2, int ca, const float *wp, void *y, int B, int Cin, int Cout, int H, int W,
                          int Ho, int Wo, int KS, int stride, int pad, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !wp || !y || H < 1 || W < 1 || Ho < 1 || Wo < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (x2 && KS == 3 && stride == 2 && pad == 1) {      // stem3 on the matrix cores, streaming rows (stem3.hip); other shapes fall through
        const int r = stem3_fwd_rows((const uint16_t *)x, (const uint16_t *)x2, ca, wp, (uint16_t *)y, B, Cin, Cout, H, W, Ho, Wo, st);
        if (r != DFINE_E_BADARG) return r;
    }
    const uint16_t *xs = (const uint16_t *)x;
    const uint16_t *xs2 = (const uint16_t *)x2;
    uint16_t *ys = (uint16_t *)y;
    // stride-1 layers of a single input tensor: matrix-core kernel (DFINE_STEM_MFMA=0: the direct kernel)
#define STEM_MFMA_CASE(CI, CO, K)                                                                                                   \
    if (!xs2 && stride == 1 && Cin == CI && Cout == CO && KS == K) {                                                       \
        constexpr int cp = (CI + 7) / 8 * 8;                                                                                        \
        const size_t lds = (size_t)(kSmRows + K - 1) * ((kSmCols + K - 1) * cp + (kSmCols + K - 1 + 7) / 8 * 8) * 2;                                                  \
        const int tx = (Wo + kSmCols - 1) / kSmCols, ty = (Ho + kSmRows - 1) / kSmRows, total = B * tx * ty;                        \
        static int per_cu = 0;                                                                                                      \
        if (!per_cu) {                                                                                                              \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stem_mfma_s1_kernel<CI, CO, K>, kStemThreads, lds) != hipSuccess || per_cu < 1) \
                per_cu = 1;                                                                                                         \
        }                                                                                                                           \
        const int resident = per_cu * 256;                       /* persistent workgroups: exactly one resident set (no second, thin round) */ \
        const int blocks = total < resident ? total : resident;                                                                     \
        hipLaunchKernelGGL((stem_mfma_s1_kernel<CI, CO, K>), dim3(blocks), dim3(kStemThreads), lds, st, xs, wp, ys, H, W, Ho, Wo, pad, \
                           tx, ty, total);                                                                                          \
        return check_launch();                                                                                                      \
    }
    // (the 2x2 layers; the 1x1 layer stem4 measured slower than the direct kernel: 62 vs 57 us and 30 vs 26 us)
    STEM_MFMA_CASE(24, 12, 2) STEM_MFMA_CASE(12, 24, 2) STEM_MFMA_CASE(16, 8, 2) STEM_MFMA_CASE(8, 16, 2)
    STEM_MFMA_CASE(32, 16, 2) STEM_MFMA_CASE(16, 32, 2)
#undef STEM_MFMA_CASE
    // 3x3 / stride 2 / pad 1 on whole 16-byte vectors (DFINE_STEM_VEC=0: the direct kernel)
#define STEM_VEC_CASE(CI, CO)                                                                                                       \
    if (KS == 3 && stride == 2 && pad == 1 && Cin == CI && Cout == CO && W % 8 == 0 && Wo % 4 == 0 && W == 2 * Wo && H == 2 * Ho) { \
        dim3 gridv((Ho * (Wo / 4) + kStemThreads - 1) / kStemThreads, B);                                                           \
        hipLaunchKernelGGL((stem_conv_s2_vec_kernel<CI, CO>), gridv, dim3(kStemThreads), 0, st, xs, xs2, ca, wp, ys, H, W, Ho, Wo);   \
        return check_launch();                                                                                                      \
    }
    // (stem1 only: 86 -> 65 us.  The 48 -> 24 layer measured SLOWER in this form - 351 vs 247 us: 3 200 long-running waves of 140
    // registers, and the packed FMAs with a scalar weight operand did not issue faster than the unpacked ones)
    STEM_VEC_CASE(3, 24) STEM_VEC_CASE(3, 16) STEM_VEC_CASE(3, 32)
#undef STEM_VEC_CASE
#define STEM_CASE(CI, CO, K, S_)                                                         \
    if (Cin == CI && Cout == CO && KS == K && stride == S_) {                            \
        launch_stem<CI, CO, K, S_>(xs, xs2, ca, wp, ys, B, H, W, Ho, Wo, pad, st);       \
        return check_launch();                                                           \
    }
    // B2 (D-FINE-m): 3->24, 24->12, 12->24, 48->24, 24->32 and the stride-1 data gradients (channels swapped)
    STEM_CASE(3, 24, 3, 2) STEM_CASE(24, 12, 2, 1) STEM_CASE(12, 24, 2, 1) STEM_CASE(48, 24, 3, 2) STEM_CASE(24, 32, 1, 1)
    STEM_CASE(32, 24, 1, 1)
    // B0 (n / s): 3->16, 16->8, 8->16, 32->16, 16->16
    STEM_CASE(3, 16, 3, 2) STEM_CASE(16, 8, 2, 1) STEM_CASE(8, 16, 2, 1) STEM_CASE(32, 16, 3, 2) STEM_CASE(16, 16, 1, 1)
    // B4 / B5 (l / x): 3->32, 32->16, 16->32, 64->32, 32->48 / 32->64
    STEM_CASE(3, 32, 3, 2) STEM_CASE(32, 16, 2, 1) STEM_CASE(16, 32, 2, 1) STEM_CASE(64, 32, 3, 2) STEM_CASE(32, 48, 1, 1)
    STEM_CASE(48, 32, 1, 1) STEM_CASE(32, 64, 1, 1) STEM_CASE(64, 32, 1, 1)
#undef STEM_CASE
    return DFINE_E_BADARG;
}

// Data gradient of a 3x3 / stride 2 / pad 1 layer: dy [B,Cout,Ho,Wo] -> dx [B,Cin,2*Ho,2*Wo]; wq from
// dfine_stem_pack_weights(mode 2).
static int stem_dgrad_impl(const void *dy, const float *wq, void *dx, void *dx2, int ca, int B, int Cin, int Cout, int Ho, int Wo,
                           void *stream);

int dfine_stem_dgrad_s2_bf16(const void *dy, const float *wq, void *dx, int B, int Cin, int Cout, int Ho, int Wo,
                             void *stream) {
    return stem_dgrad_impl(dy, wq, dx, nullptr, Cin, B, Cin, Cout, Ho, Wo, stream);
}

// The same writing the gradient of a two-tensor input (dfine_stem_conv2_bf16) as two contiguous tensors.
int dfine_stem_dgrad_s2_2_bf16(const void *dy, const float *wq, void *dxa, void *dxb, int Ca, int B, int Cin, int Cout, int Ho,
                               int Wo, void *stream) {
    if (!dxb || Ca < 1 || Ca >= Cin) return DFINE_E_BADARG;
    return stem_dgrad_impl(dy, wq, dxa, dxb, Ca, B, Cin, Cout, Ho, Wo, stream);
}

static int stem_dgrad_impl(const void *dy, const float *wq, void *dx, void *dx2, int ca, int B, int Cin, int Cout, int Ho, int Wo,
                           void *stream) {
    if (B == 0) return DFINE_OK;
    if (!dy || !wq || !dx || Ho < 1 || Wo < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (dx2) {
        const int r = stem3_bwd_rows((const uint16_t *)dy, wq, (uint16_t *)dx, (uint16_t *)dx2, ca, B, Cin, Cout, Ho, Wo, st);
        if (r != DFINE_E_BADARG) return r;
    }
    const int bt = Wo >= kStemThreads ? kStemThreads : (Wo + 63) / 64 * 64;       // e.g. Wo = 160 -> 192 threads
    dim3 grid((Wo + bt - 1) / bt, 2 * Ho, B);
#define STEM_DG(CI, CO)                                                                                         \
    if (Cin == CI && Cout == CO) {                                                                              \
        hipLaunchKernelGGL((stem_dgrad_s2_kernel<CI, CO>), grid, dim3(bt), 0, st, (const uint16_t *)dy, wq, \
                           (uint16_t *)dx, (uint16_t *)dx2, ca, 2 * Ho, 2 * Wo, Ho, Wo);                        \
        return check_launch();                                                                                  \
    }
    STEM_DG(48, 24) STEM_DG(32, 16) STEM_DG(64, 32)
#undef STEM_DG
    return DFINE_E_BADARG;
}

// the (pair group, kx) kernel serves the 3x3 / stride-2 / pad-1 and the 2x2 / stride-1 / bottom-right-padded layers of StemBlock
static bool stem_wgrad_kx_ok(int KS, int stride, int pad) {
    static const int on = [] { const char *e = getenv("DFINE_STEM_WGRAD_KX"); return e ? atoi(e) : 1; }();
    return on && ((KS == 3 && stride == 2 && pad == 1) || (KS == 2 && stride == 1 && pad == 0));
}

static int stem_wgrad_pg() {
    constexpr int pg = 3;
    return pg;
}

static void stem_wgrad_plan(int B, int Cin, int KS, int Ho, int Wo, int *ngroups, int *nsplit, int *steps, bool kx = false) {
    const int ntile = (Cin * KS * KS + 15) / 16;
    *ngroups = kx ? (Cin * KS + 16 * stem_wgrad_pg() - 1) / (16 * stem_wgrad_pg()) : (ntile + kStemNT - 1) / kStemNT;
    const int total = B * Ho * (Wo / 32);
    int sp = 6144 / *ngroups;                          // ~6000 workers = 24 waves per CU
    const int cap = (int)(32000000 / ((int64_t)Cin * KS * KS * 32 * 4)) + 1;   // fp32 partials <= ~32 MB
    if (sp > cap) sp = cap;
    if (sp > total) sp = total;
    if (sp < 1) sp = 1;
    *steps = (total + sp - 1) / sp;
    *nsplit = (total + *steps - 1) / *steps;
}

int64_t dfine_stem_wgrad_ws_floats(int B, int Cin, int Cout, int KS, int Ho, int Wo) {
    int ng, ns, st, ns2 = 0;
    stem_wgrad_plan(B, Cin, KS, Ho, Wo, &ng, &ns, &st);
    if (KS == 2 || KS == 3) stem_wgrad_plan(B, Cin, KS, Ho, Wo, &ng, &ns2, &st, true);      // (whichever kernel the launch picks)
    return (int64_t)(ns > ns2 ? ns : ns2) * Cout * Cin * KS * KS;
}

// dw [Cout,Cin,KS,KS] f32 (overwritten).  Wo % 32 == 0, Cout <= 32.
static int stem_wgrad_impl(const void *x, const void *x2, int ca, const void *dy, float *dw, float *ws, int B, int Cin, int Cout,
                           int H, int W, int Ho, int Wo, int KS, int stride, int pad, void *stream);

int dfine_stem_wgrad_bf16(const void *x, const void *dy, float *dw, float *ws, int B, int Cin, int Cout, int H, int W,
                          int Ho, int Wo, int KS, int stride, int pad, void *stream) {
    return stem_wgrad_impl(x, nullptr, Cin, dy, dw, ws, B, Cin, Cout, H, W, Ho, Wo, KS, stride, pad, stream);
}

int dfine_stem_wgrad2_bf16(const void *xa, const void *xb, int Ca, const void *dy, float *dw, float *ws, int B, int Cin, int Cout,
                           int H, int W, int Ho, int Wo, int KS, int stride, int pad, void *stream) {
    if (!xb || Ca < 1 || Ca >= Cin) return DFINE_E_BADARG;
    return stem_wgrad_impl(xa, xb, Ca, dy, dw, ws, B, Cin, Cout, H, W, Ho, Wo, KS, stride, pad, stream);
}

static int stem_wgrad_impl(const void *x, const void *x2, int ca, const void *dy, float *dw, float *ws, int B, int Cin, int Cout,
                           int H, int W, int Ho, int Wo, int KS, int stride, int pad, void *stream) {
    if (B == 0) return DFINE_OK;
    if (!x || !dy || !dw || !ws || Cout > 32 || Cout < 1 || Cin < 1 || Wo % 32 || KS < 1 || KS > 3 || pad > 1 ||
        (stride != 1 && stride != 2) || W < 8 * stride)
        return DFINE_E_BADARG;
    // one 8-output lane group reads ONE clamped x window (base = min(xo0 * S, W - 8 * S)) and shifts by kx - pad in {-1, 0, 1}:
    // a row width that is not a multiple of 8 * stride would pair the last partly valid group with shifted columns, and
    // KS = 3 without padding needs a shift of +2 - both would be silently wrong weight gradients, so they are refused.
    if (W % (8 * stride) != 0 || KS - 1 - pad > 1) return DFINE_E_BADARG;
    int ng, ns, steps;
    const bool kx = stem_wgrad_kx_ok(KS, stride, pad);
    stem_wgrad_plan(B, Cin, KS, Ho, Wo, &ng, &ns, &steps, kx);
    const int workers = ng * ns;
    hipStream_t st = (hipStream_t)stream;
#define STEM_WKX(SS, KK, PP, GG) hipLaunchKernelGGL((stem_wgrad_kx_kernel<SS, KK, PP, GG>), dim3((workers + 3) / 4), dim3(kStemThreads), 0, st, \
        (const uint16_t *)x, (const uint16_t *)x2, ca, (const uint16_t *)dy, ws, Cin, Cout, H, W, Ho, Wo, ng, ns, B * Ho * (Wo / 32), steps)
    if (kx && KS == 3) { if (stem_wgrad_pg() == 2) STEM_WKX(2, 3, 1, 2); else STEM_WKX(2, 3, 1, 3); }
    else if (kx) { if (stem_wgrad_pg() == 2) STEM_WKX(1, 2, 0, 2); else STEM_WKX(1, 2, 0, 3); }
#undef STEM_WKX
    else if (stride == 1)
        hipLaunchKernelGGL(stem_wgrad_kernel<1>, dim3((workers + 3) / 4), dim3(kStemThreads), 0, st, (const uint16_t *)x,
                           (const uint16_t *)x2, ca, (const uint16_t *)dy, ws, Cin, Cout, KS, pad, H, W, Ho, Wo, ng, ns, B * Ho * (Wo / 32), steps);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<2>, dim3((workers + 3) / 4), dim3(kStemThreads), 0, st, (const uint16_t *)x,
                           (const uint16_t *)x2, ca, (const uint16_t *)dy, ws, Cin, Cout, KS, pad, H, W, Ho, Wo, ng, ns, B * Ho * (Wo / 32), steps);
    if (int e = check_launch()) return e;
    const int total = Cout * Cin * KS * KS;
    zero_fill_async(dw, sizeof(float) * (size_t)total, st);
    const int colblocks = (total + 63) / 64;
    int chunks = 1024 / colblocks;                       // ~1024 blocks whatever the weight size
    if (chunks < 1) chunks = 1;
    if (chunks > ns) chunks = ns;
    const int spb = (ns + chunks - 1) / chunks;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(colblocks, (ns + spb - 1) / spb), dim3(256), 0, st, ws, dw, ns, total, spb);
    return check_launch();
}

int dfine_stem_pool_fwd(const void *x, void *y, int64_t planes, int H, int W, void *stream) {
    if (planes == 0) return DFINE_OK;
    if (!x || !y || H < 1 || W < 1) return DFINE_E_BADARG;
    const int64_t n = planes * H * W;
    if (W % 8 == 0)
        hipLaunchKernelGGL(stem_pool_fwd8_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (uint16_t *)y, H, W, planes);
    else
        hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (uint16_t *)y, H, W, planes);
    return check_launch();
}

// dx += the pool's gradient (W % 8 == 0 only): the gradient fan-in of stem1's output (pool + stem2a, ref hgnetv2.py:158-163) without
// the element-wise add
int dfine_stem_pool_bwd_acc(const void *x, const void *dy, void *dx, int64_t planes, int H, int W, void *stream) {
    if (planes == 0) return DFINE_OK;
    if (!x || !dy || !dx || H < 1 || W < 1 || W % 8) return DFINE_E_BADARG;
    const int64_t n = planes * H * W;
    hipLaunchKernelGGL(stem_pool_bwd8_kernel<true>, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)x, (const uint16_t *)dy, (uint16_t *)dx, H, W, planes);
    return check_launch();
}

int dfine_stem_pool_bwd(const void *x, const void *dy, void *dx, int64_t planes, int H, int W, void *stream) {
    if (planes == 0) return DFINE_OK;
    if (!x || !dy || !dx || H < 1 || W < 1) return DFINE_E_BADARG;
    const int64_t n = planes * H * W;
    if (W % 8 == 0)
        hipLaunchKernelGGL(stem_pool_bwd8_kernel<false>, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (const uint16_t *)dy, (uint16_t *)dx, H, W, planes);
    else
        hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (const uint16_t *)dy, (uint16_t *)dx, H, W, planes);
    return check_launch();
}

}  // extern "C"
