// 3x3 convolution (stride 1, "same" padding), forward / data gradient, for layers with <= 32 input AND <= 32 output channels
// on wide maps (HGNetv2 stage 1: 32 -> 32 @ 160 x 160 at a 640 x 640 input; reference: the nn.Conv2d of ConvBNAct,
// /root/reference/src/d_fine/arch/hgnetv2.py:35-80).
//
// These layers are HBM-bound (105 MB in + out, 15 GFLOP) and ran on the first-generation implicit-GEMM kernel
// (conv_igemm_kernel<3, 1, 8, 1>, conv.hip): one workgroup per (image, ROW) that loads and transposes THREE input rows for
// one output row (3 x read amplification), 64-output-channel tiles half empty - 82 us stand-alone, 108 us in the step, for
// ~19 us of HBM time (8 launches per D-FINE-m step).  Here a workgroup walks down the rows of one image:
//   * every input row is loaded ONCE, transposed in registers (8 pixels x 8 channels per lane: 8 coalesced 16-byte loads,
//     32 v_perm_b32, 8 ds_write_b128) into [pixel][32 channel] records of an LDS ring of four rows (+ one zero record at each
//     row end, zero rows above / below the image); the next-but-one row's loads are in flight during the MFMAs;
//   * a tap (kr, kc) is a record offset: B fragment of a 16-pixel tile = one ds_read_b128 per lane at record px + kc of row
//     r + kr - 1 (a wave reads 1 KiB of consecutive LDS: conflict-free); the 9 x 2 A fragments (32 x 32 weights per tap) stay
//     in registers for the life of the workgroup; 18 v_mfma_f32_16x16x32_bf16 per 16-pixel tile;
//   * the output tile goes through a per-wave LDS tile so that a lane stores 8 pixels (16 bytes) of one channel.
// The data gradient is the same kernel on the weights packed with dgrad = 1 (channels exchanged, taps flipped).
#include "common.h"

namespace dfine {

typedef __attribute__((ext_vector_type(8))) __bf16 c3_bf16x8;
typedef __attribute__((ext_vector_type(4))) float c3_f32x4;

struct C3sArgs {
    const uint16_t *x, *w2;
    uint16_t *y;
    int Cin, Cout, NP, KP, H, W, rpb, spi, accum;
};

constexpr int kC3Threads = 256;

__global__ __launch_bounds__(kC3Threads) void conv3x3_rows32_kernel(const C3sArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin, Cout = a.Cout;
    const int b = blockIdx.x / a.spi, ra = (blockIdx.x - b * a.spi) * a.rpb, rb = min(H, ra + a.rpb);
    const int RECS = W + 2, SLOT = RECS * 64;                           // records (64 B) per row image: halo, W pixels, halo
    constexpr int OP = 24;                                              // output staging pitch (elements): 16 pixels + 8
    uint16_t *ot = reinterpret_cast<uint16_t *>(lds + 4 * SLOT) + wave * (32 * OP);
    const uint16_t *xb = a.x + (int64_t)b * Cin * HW;
    uint16_t *yb = a.y + (int64_t)b * Cout * HW;

    // halo records (never written again) and the rows outside the image start as zeros
    for (int i = tid * 16; i < 4 * SLOT; i += kC3Threads * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);

    // ---- A fragments: weights [tap][NP][KP = 32]: lane -> n = nt * 16 + i16, k = 8 g .. 8 g + 7 ------------------------------
    uint4 aw[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 16 + i16;
            aw[t][nt] = n < a.NP ? *reinterpret_cast<const uint4 *>(a.w2 + ((int64_t)t * a.NP + n) * a.KP + 8 * g) : make_uint4(0, 0, 0, 0);
        }

    // ---- staging: task = (channel group cg of 8, chunk of 8 pixels); W / 8 * 4 tasks per row, one per thread ----------------
    const int ntask = (W >> 3) * 4;
    const int cg = tid & 3, chunk = tid >> 2;
    const bool has_task = tid < ntask;
    uint4 pf[8];
    auto fetch = [&](int grow) {                                        // global row -> registers (zeros outside the image / layer)
        const bool rowok = has_task && grow >= 0 && grow < H;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = cg * 8 + c;
            pf[c] = make_uint4(0, 0, 0, 0);
            if (rowok && ch < Cin) pf[c] = *reinterpret_cast<const uint4 *>(xb + ((int64_t)ch * H + grow) * W + chunk * 8);
        }
    };
    auto commit = [&](int slot) {                                       // 8 x 8 transpose in registers, one record slice per pixel
        if (!has_task) return;
        unsigned char *d = lds + slot * SLOT + (chunk * 8 + 1) * 64 + cg * 16;
        const uint32_t(*w)[4] = reinterpret_cast<const uint32_t(*)[4]>(&pf[0]);          // w[channel][pixel pair]
#pragma unroll
        for (int pix = 0; pix < 8; ++pix) {
            const uint32_t sel = (pix & 1) ? 0x07060302u : 0x05040100u;
            const int dw = pix >> 1;
            uint4 o;
            o.x = __builtin_amdgcn_perm(w[1][dw], w[0][dw], sel);
            o.y = __builtin_amdgcn_perm(w[3][dw], w[2][dw], sel);
            o.z = __builtin_amdgcn_perm(w[5][dw], w[4][dw], sel);
            o.w = __builtin_amdgcn_perm(w[7][dw], w[6][dw], sel);
            *reinterpret_cast<uint4 *>(d + pix * 64) = o;
        }
    };
    // row `grow` lives in slot (grow + 1) & 3 (grow >= -1)
    __syncthreads();                                                    // zero fill done
    fetch(ra - 1); commit((ra) & 3);
    fetch(ra); commit((ra + 1) & 3);
    fetch(ra + 1);                                                      // committed at the top of the first iteration
    const int npt = W >> 4;
    for (int r = ra; r < rb; ++r) {
        commit((r + 2) & 3);                                            // row r + 1 (its slot held row r - 3: free since the last barrier)
        __syncthreads();
        fetch(r + 2);                                                   // in flight during the MFMAs below
        const unsigned char *row0 = lds + ((r + 0) & 3) * SLOT, *row1 = lds + ((r + 1) & 3) * SLOT, *row2 = lds + ((r + 2) & 3) * SLOT;
        for (int t = wave; t < npt; t += 4) {
            const int rec = (t * 16 + i16) * 64 + g * 16;              // tap kc adds kc records (record 0 = left halo)
            c3_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            uint4 bf[9];
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                bf[0 + kc] = *reinterpret_cast<const uint4 *>(row0 + rec + kc * 64);
                bf[3 + kc] = *reinterpret_cast<const uint4 *>(row1 + rec + kc * 64);
                bf[6 + kc] = *reinterpret_cast<const uint4 *>(row2 + rec + kc * 64);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const c3_bf16x8 bv = __builtin_bit_cast(c3_bf16x8, bf[tap]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c3_bf16x8, aw[tap][0]), bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c3_bf16x8, aw[tap][1]), bv, acc1, 0, 0, 0);
            }
            // D: lane -> n = 4 g + reg (+ 16), pixel = i16.  Through the wave's LDS tile: a lane then owns 8 pixels of one channel.
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ot[(4 * g + q) * OP + i16] = f32_to_bf16(acc0[q]);
                ot[(16 + 4 * g + q) * OP + i16] = f32_to_bf16(acc1[q]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int n = lane >> 1, half = lane & 1;
            if (n < Cout) {
                uint16_t *yp = yb + ((int64_t)n * H + r) * W + t * 16 + half * 8;
                uint4 v = *reinterpret_cast<const uint4 *>(ot + n * OP + half * 8);
                if (a.accum) {                                          // y += conv(x): one rounding, like a separate add
                    const uint4 o = *reinterpret_cast<const uint4 *>(yp);
                    const uint32_t av[4] = {v.x, v.y, v.z, v.w}, cv[4] = {o.x, o.y, o.z, o.w};
                    uint32_t rr[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        rr[k] = pack_bf16x2(__uint_as_float(av[k] << 16) + __uint_as_float(cv[k] << 16),
                                            __uint_as_float(av[k] & 0xffff0000u) + __uint_as_float(cv[k] & 0xffff0000u));
                    v = make_uint4(rr[0], rr[1], rr[2], rr[3]);
                }
                *reinterpret_cast<uint4 *>(yp) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile's reads are done before the next tile's writes
        }
    }
}

// Shapes the kernel takes: KP == 32 (<= 32 input channels), NP <= 32, W a multiple of 16 with one staging task per thread.
bool conv3x3_rows32_ok(int NP, int KP, int H, int W) {
    static const int on = [] { const char *e = getenv("DFINE_CONV3X3_ROWS32"); return e ? atoi(e) : 1; }();
    return on && KP == 32 && NP <= 32 && W % 16 == 0 && W >= 32 && (W / 8) * 4 <= kC3Threads && H >= 1;
}

int conv3x3_rows32_launch(const uint16_t *x, const uint16_t *w2, uint16_t *y, int B, int Cin, int Cout, int NP, int KP, int H, int W,
                          int accum, hipStream_t st) {
    if (!conv3x3_rows32_ok(NP, KP, H, W)) return DFINE_E_BADARG;
    C3sArgs a;
    a.x = x; a.w2 = w2; a.y = y; a.Cin = Cin; a.Cout = Cout; a.NP = NP; a.KP = KP; a.H = H; a.W = W; a.accum = accum;
    // rows per workgroup: two halo rows are re-read per workgroup, ~2 workgroups per CU keep each other's load / MFMA phases covered
    int spi = (512 + B - 1) / B;
    if (spi > (H + 7) / 8) spi = (H + 7) / 8;                          // at least 8 rows each
    if (spi < 1) spi = 1;
    a.rpb = (H + spi - 1) / spi;
    a.spi = (H + a.rpb - 1) / a.rpb;
    const size_t ldsb = (size_t)4 * (W + 2) * 64 + 4 * 32 * 24 * 2;
    static bool attr_set = false;                     // once: not a stream operation, keep it out of graph capture
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_rows32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_rows32_kernel, dim3(B * a.spi), dim3(kC3Threads), ldsb, st, a);
    return check_launch();
}

}  // namespace dfine
