// HGNetv2 stem3 - the 3x3 / stride 2 / pad 1 convolution over the concatenation [max-pooled stem1 | stem2b]
// (src/d_fine/arch/hgnetv2.py:115-166: 48 -> 24 channels, 320 x 320 -> 160 x 160 at a 640 x 640 input; B0: 32 -> 16, B4 / B5: 64 -> 32) -
// forward and data gradient on the matrix cores, streaming rows.
//
// Both are bandwidth layers (354 MB of bf16 activations, 17 GFLOP at batch 32), and ran on the direct kernels of stem.hip: one
// thread per output pixel (pair) with fp32 FMAs on 2-byte loads - bound by the vector unit (254 us forward, 385 us backward for
// ~65 us of HBM time each).  Here, as in conv3s.hip, a workgroup walks down a band of rows:
//   forward   every input row of the band is loaded ONCE (16-byte loads along the row, 8 x 8 transpose in registers) into
//             [pixel][channel] records of an LDS ring of four rows; output pixel o, tap (kr, kc) = record 2 o + kc of row
//             2 r + kr - 1: one ds_read_b128 (+ one ds_read_b64 for the 16-channel tail of 48) per B fragment; the weight
//             fragments stay in registers; 16x16x32 (+ 16x16x16) MFMAs; the output leaves through a per-wave LDS tile as 16-byte
//             stores.  A workgroup owns 80 output columns (160 input columns + one halo column): 62 KiB of LDS, two per CU.
//   backward  dx rows 2 m and 2 m + 1 from dy rows m and m + 1: the taps of a stride-2 transposed convolution depend only on the
//             parity of the dx pixel - (even row: ky = 1 | odd row: ky = 0 of dy row m + 1 and ky = 2 of row m) x (even column:
//             kx = 1 | odd column: kx = 0 of dy column xo + 1 and kx = 2 of xo) - so a tile of 16 dy columns gives 32 dx columns
//             of both rows with 27 MFMAs per 16 input channels; dy rows live in LDS as [pixel][32-channel] records (zero padded).
// Weights: the fp32 arrays of dfine_stem_pack_weights (mode 0 forward, mode 2 backward), rounded to bf16 fragments once per
// workgroup - bf16 weights, fp32 accumulation, like every other convolution under bf16 autocast.
#include <cstdlib>

#include "common.h"

namespace dfine {

typedef __attribute__((ext_vector_type(8))) __bf16 s3_bf16x8;
typedef __attribute__((ext_vector_type(4))) short s3_s16x4;
typedef __attribute__((ext_vector_type(4))) float s3_f32x4;

constexpr int kS3Threads = 256;
constexpr int kS3OutCols = 80;               // output columns per workgroup (forward)
#ifndef DFINE_S3_PAD
#define DFINE_S3_PAD 16
#endif
constexpr int kS3Pad = DFINE_S3_PAD;

// 8 x 8 transpose of 8 channel rows (16 bytes = 8 pixels each) into 8 pixel vectors of 8 channels
__device__ __forceinline__ void s3_transpose8(const uint4 (&pf)[8], uint4 (&o)[8]) {
    const uint32_t(*w)[4] = reinterpret_cast<const uint32_t(*)[4]>(&pf[0]);          // w[channel][pixel pair]
#pragma unroll
    for (int pix = 0; pix < 8; ++pix) {
        const uint32_t sel = (pix & 1) ? 0x07060302u : 0x05040100u;
        const int dw = pix >> 1;
        o[pix].x = __builtin_amdgcn_perm(w[1][dw], w[0][dw], sel);
        o[pix].y = __builtin_amdgcn_perm(w[3][dw], w[2][dw], sel);
        o[pix].z = __builtin_amdgcn_perm(w[5][dw], w[4][dw], sel);
        o[pix].w = __builtin_amdgcn_perm(w[7][dw], w[6][dw], sel);
    }
}

struct S3FwdArgs {
    const uint16_t *xa, *xb;     // [B, CIN / 2, H, W] each
    const float *wp;             // [(ci * 3 + ky) * 3 + kx][COUT]
    uint16_t *y;                 // [B, COUT, Ho, Wo]
    int H, W, Ho, Wo, rpb, bands, cblocks;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(kS3Threads) void stem3_fwd_rows_kernel(const S3FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int K32 = CIN / 32, K16 = (CIN % 32) / 16, NT = (COUT + 15) / 16, CG = CIN / 8, CH = CIN / 2;
    constexpr int RB = CIN * 2;                                        // bytes per pixel record
    // record 0 = the column left of the block.  16 bytes of padding behind every 8 records: the staging writes of a wave go to the
    // same pixel of DIFFERENT 8-pixel chunks, 8 records = 768 bytes apart - an even number of 16-byte units, i.e. the same banks
    constexpr int IW = 2 * kS3OutCols, REC = IW + 1, SLOT = REC * RB + (REC / 8 + 1) * kS3Pad;
    auto recoff = [](int rec) -> int { return rec * RB + (rec >> 3) * kS3Pad; };
    constexpr int NCHUNK = IW / 8 + 1;                                 // 8-pixel chunks per row, chunk 0 = the halo's
    constexpr int OP = 24;                                             // output staging pitch (elements)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    int blk = blockIdx.x;
    const int cb = blk % a.cblocks; blk /= a.cblocks;
    const int band = blk % a.bands, b = blk / a.bands;
    const int ra = band * a.rpb, rb = min(Ho, ra + a.rpb);
    const int c0 = cb * IW;                                            // first input column of the block
    uint16_t *ot = reinterpret_cast<uint16_t *>(lds + 4 * SLOT) + wave * (16 * NT * OP);
    const uint16_t *xab = a.xa + (int64_t)b * CH * H * W, *xbb = a.xb + (int64_t)b * CH * H * W;

    // ---- weight fragments: lane -> output channel nt * 16 + i16, input channels 8 g .. 8 g + 7 of slab s (4 g .. 4 g + 3 of the tail)
    uint4 aw[9][K32 > 0 ? K32 : 1][NT];
    uint2 at[9][K16 > 0 ? K16 : 1][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + i16, coc = co < COUT ? co : COUT - 1;      // (clamped loads + select: no divergent branches)
            const bool cok = co < COUT;
#pragma unroll
            for (int s = 0; s < K32; ++s) {
                uint32_t p[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = s * 32 + 8 * g + 2 * e;
                    const float w0 = a.wp[(ci * 9 + t) * COUT + coc], w1 = a.wp[((ci + 1) * 9 + t) * COUT + coc];
                    p[e] = cok ? pack_bf16x2(w0, w1) : 0u;
                }
                aw[t][s][nt] = make_uint4(p[0], p[1], p[2], p[3]);
            }
#pragma unroll
            for (int s = 0; s < K16; ++s) {
                uint32_t p[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ci = K32 * 32 + s * 16 + 4 * g + 2 * e;
                    const float w0 = a.wp[(ci * 9 + t) * COUT + coc], w1 = a.wp[((ci + 1) * 9 + t) * COUT + coc];
                    p[e] = cok ? pack_bf16x2(w0, w1) : 0u;
                }
                at[t][s][nt] = make_uint2(p[0], p[1]);
            }
        }

    // ---- staging: task = (row of the pair, channel group of 8, 8-pixel chunk); input row j lives in slot (j + 1) & 3 ----------
    constexpr int TPR = NCHUNK * CG;                                   // tasks per row
    constexpr int NTK = (2 * TPR + kS3Threads - 1) / kS3Threads;       // tasks per thread and row pair (1; 2 for 64 channels)
    uint4 pf[NTK][8];
    auto fetch = [&](int pair) {                                       // input rows 2 pair - 1 (trow 0) and 2 pair (trow 1)
#pragma unroll
        for (int k = 0; k < NTK; ++k) {
            const int task = tid + k * kS3Threads;
            const int trow = task / TPR, tt = task - trow * TPR, cg = tt % CG, chunk = tt / CG;   // chunk 0: columns c0 - 8 .. c0 - 1
            const int grow = 2 * pair - 1 + trow, col = c0 + (chunk - 1) * 8;
            const bool ok = task < 2 * TPR && grow >= 0 && grow < H && col >= 0 && col < W;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ch = cg * 8 + c;
                const uint16_t *src = ch < CH ? xab + (int64_t)ch * H * W : xbb + (int64_t)(ch - CH) * H * W;
                pf[k][c] = make_uint4(0, 0, 0, 0);
                if (ok) pf[k][c] = *reinterpret_cast<const uint4 *>(src + (int64_t)grow * W + col);
            }
        }
    };
    auto commit = [&](int pair) {
#pragma unroll
        for (int k = 0; k < NTK; ++k) {
            const int task = tid + k * kS3Threads;
            if (task >= 2 * TPR) continue;
            const int trow = task / TPR, tt = task - trow * TPR, cg = tt % CG, chunk = tt / CG;
            uint4 o[8];
            s3_transpose8(pf[k], o);
            const int grow = 2 * pair - 1 + trow;
            unsigned char *d = lds + ((grow + 1) & 3) * SLOT + cg * 16;
            if (chunk == 0) {
                *reinterpret_cast<uint4 *>(d) = o[7];                  // record 0 = column c0 - 1 (only the last pixel of the chunk is kept)
            } else {
#pragma unroll
                for (int pix = 0; pix < 8; ++pix) *reinterpret_cast<uint4 *>(d + recoff(1 + (chunk - 1) * 8 + pix)) = o[pix];
            }
        }
    };

    fetch(ra); commit(ra);
    fetch(ra + 1);
    constexpr int NPT = kS3OutCols / 16;
    for (int r = ra; r < rb; ++r) {
        commit(r + 1);                                                 // rows 2 r + 1, 2 r + 2 (their slots held rows 2 r - 3, 2 r - 2)
        __syncthreads();
        fetch(r + 2);                                                  // in flight during the MFMAs below
        const unsigned char *rows[3] = {lds + ((2 * r) & 3) * SLOT, lds + ((2 * r + 1) & 3) * SLOT, lds + ((2 * r + 2) & 3) * SLOT};
        for (int t = wave; t < NPT; t += 4) {
            // The 16x16x16 products of the 16-channel tail go to accumulators of their own: chained onto the 16x16x32 ones (same
            // srcC / vDst) the last 16x16x32 result came out with rows 4 g, 4 g + 1 of every tile lost - two MFMA shapes back to
            // back on one accumulator is a hazard this toolchain does not pad for gfx950 (found by tests/test_stem_gpu.py).
            s3_f32x4 acc[NT], acc2[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc[nt] = s3_f32x4{0.f, 0.f, 0.f, 0.f}; acc2[nt] = s3_f32x4{0.f, 0.f, 0.f, 0.f}; }
            const int rec0 = 2 * (t * 16 + i16);
            const int roff[3] = {recoff(rec0), recoff(rec0 + 1), recoff(rec0 + 2)};
#pragma unroll
            for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                for (int kc = 0; kc < 3; ++kc) {
                    const unsigned char *p = rows[kr] + roff[kc];
#pragma unroll
                    for (int s = 0; s < K32; ++s) {
                        const s3_bf16x8 bv = __builtin_bit_cast(s3_bf16x8, *reinterpret_cast<const uint4 *>(p + s * 64 + g * 16));
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s3_bf16x8, aw[kr * 3 + kc][s][nt]), bv, acc[nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < K16; ++s) {
                        const s3_s16x4 bv = __builtin_bit_cast(s3_s16x4, *reinterpret_cast<const uint2 *>(p + K32 * 64 + s * 32 + g * 8));
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s3_s16x4, at[kr * 3 + kc][s][nt]), bv, acc2[nt], 0, 0, 0);
                    }
                }
            // D: lane -> channel nt * 16 + 4 g + q, pixel i16.  Through the wave's LDS tile: a lane then owns 8 pixels of one channel.
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) ot[(nt * 16 + 4 * g + q) * OP + i16] = f32_to_bf16(K16 ? acc[nt][q] + acc2[nt][q] : acc[nt][q]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int n = lane >> 1, half = lane & 1;
            if (n < COUT)
                *reinterpret_cast<uint4 *>(a.y + (((int64_t)b * COUT + n) * Ho + r) * Wo + cb * kS3OutCols + t * 16 + half * 8) =
                    *reinterpret_cast<const uint4 *>(ot + n * OP + half * 8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile's reads are done before the next tile's writes
        }
        __syncthreads();                                               // every wave is done with rows 2 r - 1, 2 r: the next commit reuses their slots
    }
}

struct S3BwdArgs {
    const uint16_t *dy;          // [B, COUT, Ho, Wo]
    const float *wq;             // [(co * 3 + ky) * 3 + kx][CIN]
    uint16_t *dxa, *dxb;         // [B, CIN / 2, 2 Ho, 2 Wo] each
    int Ho, Wo, rpb, bands;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(kS3Threads) void stem3_bwd_rows_kernel(const S3BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = CIN / 16, CH = CIN / 2, CGO = (COUT + 7) / 8;    // dx channel tiles; dy channel groups of 8 (records hold 32)
    constexpr int OP = 40;                                              // output staging pitch: 32 dx pixels + 8
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int Ho = a.Ho, Wo = a.Wo, H = 2 * Ho, W = 2 * Wo;
    const int band = blockIdx.x % a.bands, b = blockIdx.x / a.bands;
    const int ma = band * a.rpb, mb = min(Ho, ma + a.rpb);
    const int REC = Wo + 1, SLOT = REC * 64;                           // record Wo = the (zero) column right of the row
    uint16_t *ot = reinterpret_cast<uint16_t *>(lds + 4 * SLOT) + wave * (CIN * OP);
    const uint16_t *dyb = a.dy + (int64_t)b * COUT * Ho * Wo;

    for (int i = tid * 16; i < 4 * SLOT; i += kS3Threads * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);

    // ---- weight fragments: lane -> input channel nt * 16 + i16, output channels 8 g .. 8 g + 7 (zero past COUT) -------------------
    uint4 aw[9][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ci = nt * 16 + i16;
            uint32_t p[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 8 * g + 2 * e, c0c = co < COUT ? co : COUT - 1, c1c = co + 1 < COUT ? co + 1 : COUT - 1;
                const float w0 = a.wq[(c0c * 9 + t) * CIN + ci], w1 = a.wq[(c1c * 9 + t) * CIN + ci];
                p[e] = pack_bf16x2(co < COUT ? w0 : 0.f, co + 1 < COUT ? w1 : 0.f);
            }
            aw[t][nt] = make_uint4(p[0], p[1], p[2], p[3]);
        }

    // ---- staging of dy rows: task = (channel group of 8, 8-pixel chunk); dy row j lives in slot j & 3 -----------------------------
    const int ntask = (Wo >> 3) * CGO;
    const int cg = tid % CGO, chunk = tid / CGO;
    const bool has_task = tid < ntask;
    uint4 pf[8];
    auto fetch = [&](int row) {
        const bool ok = has_task && row < Ho;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = cg * 8 + c;
            pf[c] = make_uint4(0, 0, 0, 0);
            if (ok && ch < COUT) pf[c] = *reinterpret_cast<const uint4 *>(dyb + ((int64_t)ch * Ho + row) * Wo + chunk * 8);
        }
    };
    auto commit = [&](int row) {
        if (!has_task) return;
        uint4 o[8];
        s3_transpose8(pf, o);
        unsigned char *d = lds + (row & 3) * SLOT + chunk * 8 * 64 + cg * 16;
#pragma unroll
        for (int pix = 0; pix < 8; ++pix) *reinterpret_cast<uint4 *>(d + pix * 64) = o[pix];
    };
    __syncthreads();                                                   // zero fill done (channels past COUT and the halo column stay zero)
    fetch(ma); commit(ma);
    fetch(ma + 1);
    const int npt = Wo >> 4;
    for (int m = ma; m < mb; ++m) {
        commit(m + 1);                                                 // (a row past the plane commits zeros)
        __syncthreads();
        fetch(m + 2);
        const unsigned char *r0 = lds + (m & 3) * SLOT, *r1 = lds + ((m + 1) & 3) * SLOT;
        for (int t = wave; t < npt; t += 4) {
            const int rec = (t * 16 + i16) * 64 + g * 16;
            const s3_bf16x8 b00 = __builtin_bit_cast(s3_bf16x8, *reinterpret_cast<const uint4 *>(r0 + rec));        // dy[m][xo]
            const s3_bf16x8 b01 = __builtin_bit_cast(s3_bf16x8, *reinterpret_cast<const uint4 *>(r0 + rec + 64));   // dy[m][xo + 1]
            const s3_bf16x8 b10 = __builtin_bit_cast(s3_bf16x8, *reinterpret_cast<const uint4 *>(r1 + rec));        // dy[m + 1][xo]
            const s3_bf16x8 b11 = __builtin_bit_cast(s3_bf16x8, *reinterpret_cast<const uint4 *>(r1 + rec + 64));   // dy[m + 1][xo + 1]
#pragma unroll
            for (int par = 0; par < 2; ++par) {                        // dx row 2 m + par
                s3_f32x4 ae[NT], ao[NT];                               // even / odd dx columns 2 xo, 2 xo + 1
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const s3_f32x4 z = {0.f, 0.f, 0.f, 0.f};
#define S3_W(KY, KX) __builtin_bit_cast(s3_bf16x8, aw[(KY) * 3 + (KX)][nt])
                    if (par == 0) {                                    // ky = 1, dy row m
                        ae[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(1, 1), b00, z, 0, 0, 0);
                        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(1, 0), b01, z, 0, 0, 0);
                        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(1, 2), b00, ao[nt], 0, 0, 0);
                    } else {                                           // ky = 0 of dy row m + 1, ky = 2 of dy row m
                        ae[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(0, 1), b10, z, 0, 0, 0);
                        ae[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(2, 1), b00, ae[nt], 0, 0, 0);
                        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(0, 0), b11, z, 0, 0, 0);
                        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(0, 2), b10, ao[nt], 0, 0, 0);
                        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(2, 0), b01, ao[nt], 0, 0, 0);
                        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S3_W(2, 2), b00, ao[nt], 0, 0, 0);
                    }
#undef S3_W
                }
                // D: lane -> channel nt * 16 + 4 g + q, dy column i16 -> dx columns 2 i16 (even), 2 i16 + 1 (odd): one packed pair
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint32_t *>(ot + (nt * 16 + 4 * g + q) * OP + 2 * i16) = pack_bf16x2(ae[nt][q], ao[nt][q]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int yi = 2 * m + par;
                for (int it = lane; it < CIN * 4; it += 64) {          // (channel, 8-pixel quarter of the 32 dx columns)
                    const int n = it >> 2, qd = it & 3;
                    uint16_t *dst = (n < CH ? a.dxa + ((int64_t)b * CH + n) * H * W : a.dxb + ((int64_t)b * CH + (n - CH)) * H * W) +
                                    (int64_t)yi * W + t * 32 + qd * 8;
                    *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(ot + n * OP + qd * 8);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
}

static bool stem3_rows_on() {
    static const int on = [] { const char *e = getenv("DFINE_STEM3_ROWS"); return e ? atoi(e) : 1; }();
    return on != 0;
}

template <int CIN, int COUT>
static int stem3_fwd_launch(const uint16_t *xa, const uint16_t *xb, const float *wp, uint16_t *y, int B, int H, int W, int Ho, int Wo,
                            hipStream_t st) {
    S3FwdArgs a;
    a.xa = xa; a.xb = xb; a.wp = wp; a.y = y; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.cblocks = Wo / kS3OutCols;
    constexpr int wgs = 512;
    int bands = (wgs + B * a.cblocks - 1) / (B * a.cblocks);           // ~512 workgroups: two per CU
    if (bands > (Ho + 7) / 8) bands = (Ho + 7) / 8;                    // at least 8 output rows each (one halo row pair per band)
    if (bands < 1) bands = 1;
    a.rpb = (Ho + bands - 1) / bands;
    a.bands = (Ho + a.rpb - 1) / a.rpb;
    constexpr int NT = (COUT + 15) / 16;
    const size_t ldsb = (size_t)4 * ((2 * kS3OutCols + 1) * CIN * 2 + ((2 * kS3OutCols + 1) / 8 + 1) * kS3Pad) + 4 * 16 * NT * 24 * 2;
    static bool attr_set = false;                     // once: not a stream operation, keep it out of graph capture
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stem3_fwd_rows_kernel<CIN, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL((stem3_fwd_rows_kernel<CIN, COUT>), dim3(B * a.bands * a.cblocks), dim3(kS3Threads), ldsb, st, a);
    return check_launch();
}

// xa, xb: the two halves of the input channels.  DFINE_E_BADARG: shape not served (the caller runs the direct kernel).
int stem3_fwd_rows(const uint16_t *xa, const uint16_t *xb, int Ca, const float *wp, uint16_t *y, int B, int Cin, int Cout, int H, int W,
                   int Ho, int Wo, hipStream_t st) {
    if (!stem3_rows_on() || !xb || 2 * Ca != Cin || H != 2 * Ho || W != 2 * Wo || Wo % kS3OutCols) return DFINE_E_BADARG;
    if (Cin == 48 && Cout == 24) return stem3_fwd_launch<48, 24>(xa, xb, wp, y, B, H, W, Ho, Wo, st);
    if (Cin == 32 && Cout == 16) return stem3_fwd_launch<32, 16>(xa, xb, wp, y, B, H, W, Ho, Wo, st);
    if (Cin == 64 && Cout == 32) return stem3_fwd_launch<64, 32>(xa, xb, wp, y, B, H, W, Ho, Wo, st);
    return DFINE_E_BADARG;
}

template <int CIN, int COUT>
static int stem3_bwd_launch(const uint16_t *dy, const float *wq, uint16_t *dxa, uint16_t *dxb, int B, int Ho, int Wo, hipStream_t st) {
    S3BwdArgs a;
    a.dy = dy; a.wq = wq; a.dxa = dxa; a.dxb = dxb; a.Ho = Ho; a.Wo = Wo;
    constexpr int wgs = 512;
    int bands = (wgs + B - 1) / B;
    if (bands > (Ho + 7) / 8) bands = (Ho + 7) / 8;
    if (bands < 1) bands = 1;
    a.rpb = (Ho + bands - 1) / bands;
    a.bands = (Ho + a.rpb - 1) / a.rpb;
    const size_t ldsb = (size_t)4 * (Wo + 1) * 64 + (size_t)4 * CIN * 40 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stem3_bwd_rows_kernel<CIN, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL((stem3_bwd_rows_kernel<CIN, COUT>), dim3(B * a.bands), dim3(kS3Threads), ldsb, st, a);
    return check_launch();
}

int stem3_bwd_rows(const uint16_t *dy, const float *wq, uint16_t *dxa, uint16_t *dxb, int Ca, int B, int Cin, int Cout, int Ho, int Wo,
                   hipStream_t st) {
    if (!stem3_rows_on() || !dxb || 2 * Ca != Cin || Wo % 16 || (Wo / 8) * ((Cout + 7) / 8) > kS3Threads || Wo > 480) return DFINE_E_BADARG;
    if (Cin == 48 && Cout == 24) return stem3_bwd_launch<48, 24>(dy, wq, dxa, dxb, B, Ho, Wo, st);
    if (Cin == 32 && Cout == 16) return stem3_bwd_launch<32, 16>(dy, wq, dxa, dxb, B, Ho, Wo, st);
    if (Cin == 64 && Cout == 32) return stem3_bwd_launch<64, 32>(dy, wq, dxa, dxb, B, Ho, Wo, st);
    return DFINE_E_BADARG;
}

}  // namespace dfine
