// 3x3 weight gradient, second generation (round 5): image ROWS streamed through an LDS ring by LDS-DMA, all nine taps in one
// workgroup, v_mfma_f32_32x32x16_bf16.
//
//   dW[n][c][kr][kc] = sum_{b, r, p} dY[b][n][r][p] * X[b][c][r + kr - 1][p + kc - 1]        (zero outside the map)
//
// Reference call site: the autograd of nn.Conv2d(k = 3, s = 1, p = 1) inside ConvBNAct / ConvNormLayer / VGGBlock
// (/root/reference/src/d_fine/arch/hgnetv2.py:35-80, hybrid_encoder.py:21-156), i.e. cuDNN / MIOpen's backward-weights.
//
// What bounded the first generation (conv_wgrad_kernel<3>, conv.hip: 64 x 64 (n, c) tiles, one kernel ROW per workgroup,
// register-staged strips): every workgroup re-loads its 64 + 64 channel strip for 3 of the 9 taps only, so a 128 -> 128
// layer moves each operand 2 x 3 = 6 times through the CU load path (614 MB for a 105 MB layer at 80 x 80: 5.7 TB/s of
// L2 / Infinity-Cache traffic at 107 us - the load path, not the matrix cores, set the time: 0.10-0.22 of the MFMA peak).
// Here a workgroup owns a (32 TNW) x (32 TCW) tile of (n, c) with ALL nine taps (9 x 16 accumulator registers per 32 x 32
// wave tile) and walks down the rows of one image: per output row it needs ONE new row of X (rows r - 1, r, r + 1 stay in a
// ring) and one row of dY, so an operand is loaded once per tile column / row of the weight matrix (2 x for 128 channels).
//
// LDS row image (both operands): [channel][cpr chunks of 16 B] = | zero pad | W / 8 data chunks | (zero pad) |, cpr odd so
// that the 16 lanes of a ds_read_b128 service group (16 channels, same column) fall on 16 distinct 16-byte slots; when
// W / 8 is even the right pad is the next channel's left pad.  Filled by global_load_lds_dwordx4: the LDS side is lane-linear, the per-lane GLOBAL address picks the chunk, pads
// and rows outside the image come from a 16-byte page of zeros.
//
// The +-1 COLUMN shift is applied to dY, the +-1 ROW shift to X:  with p' the column of X,
//   dW[.][.][kr][kc] += dY[row r][p' - kc + 1 ...] (x) X[row r + kr - 1][p' ...]
// so a 16-pixel K step reads three aligned X fragments (one per kernel row) and the aligned dY chunk plus its two
// neighbours, from which the two shifted dY fragments are made with five v_alignbit (the pads make the row ends right: a
// shifted-in column outside the map is a zero).  9 MFMAs (288 cycles) per 5 ds_read_b128 and ~6 vector instructions.
//
// Waves: TNW x TCW wave tiles x WK row groups (wave group k takes row r0 + k of a unit of WK rows) x WS step groups (K steps
// s = ks mod WS of the row); the WK x WS groups of a tile add their accumulators through LDS at the end.  One barrier per unit:
// counted wait for the unit's rows (PF units are in flight) -> barrier -> issue unit u + PF into the slots unit u - 1 has
// released -> MFMAs, the next K step's fragments requested ahead of the current step's MFMAs.
#include "common.h"

namespace dfine {

typedef __attribute__((ext_vector_type(8))) __bf16 w3_bf16x8;
typedef __attribute__((ext_vector_type(16))) float w3_f32x16;

__device__ uint4 g_w3_zero_page = {0u, 0u, 0u, 0u};

// see glds16 in conv.hip: inline asm so that hipcc's counters do not drain the ring in front of every ds_read
__device__ __forceinline__ void w3_glds16(const uint16_t *gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

struct W3Args {
    const uint16_t *x, *dy;
    float *part;
    int Cin, Cout, H, W, cpr, rps, spi, nct, NP16, CP16, ntiles, nsplits, lds_bytes, ablate;
};

__device__ __forceinline__ void w3_wait_vmcnt(int n) {       // s_waitcnt takes an immediate; n is wave-uniform
    switch (n) {
#define W3_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W3_VM(1) W3_VM(2) W3_VM(3) W3_VM(4) W3_VM(5) W3_VM(6) W3_VM(7) W3_VM(8) W3_VM(9) W3_VM(10) W3_VM(11) W3_VM(12)
        W3_VM(13) W3_VM(14) W3_VM(15) W3_VM(16) W3_VM(17) W3_VM(18) W3_VM(19) W3_VM(20) W3_VM(21) W3_VM(22) W3_VM(23) W3_VM(24)
#undef W3_VM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct W3Frag { uint4 ce, ri, x0, x1, x2; };

// PF units of rows in flight (loads issued PF units ahead of their MFMAs); kMaxP: bound of the LDS-DMA pieces a wave issues per unit.
// OPT (bit set; the library builds 5): 1 = partial sums leave through an LDS image of the tile with
// 16-byte stores (else straight from the accumulators: 4 bytes per lane at a 36-byte stride), 4 = early / late issue of the next
// unit's pieces by the two waves of a SIMD (see the loop).  Measured on the six 3x3 layer shapes of D-FINE-m (sum per step,
// stand-alone): OPT 0 1.477 ms, 1 1.417, 5 1.324 (first-generation kernel: 1.78).  Tried and dropped: requesting the next K step's
// fragments before the current MFMAs (two register sets: 1.548 - hipcc waits lgkmcnt(0) in front of the MFMAs anyway), issuing the
// pieces one by one behind the K steps (2.2 ms: the guarded unrolled issue code in the K loop costs more than the stalls it avoids).
template <int TNW, int TCW, int WK, int WS, int PF, int kMaxP, int OPT>
__global__ __launch_bounds__(64 * TNW * TCW * WK * WS, 2) void conv_wgrad3_rows_kernel(const W3Args a) {
    constexpr int NW = TNW * TCW * WK * WS, NT = 64 * NW, TN = 32 * TNW, TC = 32 * TCW, G = WK * WS;
    constexpr int NXR = WK + 2 + PF * WK, NDR = (PF + 1) * WK;
    static_assert((PF - 1) * kMaxP <= 24, "w3_wait_vmcnt covers 0..24");
    static_assert(WK <= 2, "row of a piece = piece >= pieces per row");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tcw = wave % TCW, tnw = (wave / TCW) % TNW, grp = wave / (TCW * TNW), k = grp / WS, ks = grp % WS;
    const int nsplits = a.nsplits, ntiles = a.ntiles;
    int tile, split;
    if ((nsplits & 7) == 0) {                    // all tiles of a split on one XCD: they share the rows of X and dY in its L2
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tile = slot % ntiles; split = (slot / ntiles) * 8 + xcd;
    } else {
        tile = blockIdx.x % ntiles; split = blockIdx.x / ntiles;
    }
    if (split >= nsplits) return;
    const int nt = tile / a.nct, ct = tile - nt * a.nct;
    const int n0 = nt * TN, c0 = ct * TC;
    const int b = split / a.spi, ra = (split - b * a.spi) * a.rps, rb = min(a.H, ra + a.rps);
    const int H = a.H, W = a.W, HW = H * W, cpr = a.cpr, CS = cpr * 16, w8 = W >> 3, ablate = a.ablate;
    const int PPRX = (TC * cpr + 63) >> 6, PPRD = (TN * cpr + 63) >> 6;            // 1 KiB pieces per row image
    const int XSLOT = PPRX * 1024, DSLOT = PPRD * 1024, XRING = NXR * XSLOT;
    const int NPX = WK * PPRX, NPU = WK * (PPRX + PPRD);                           // x pieces / all pieces per unit
    const int np_mine = (NPU - wave + NW - 1) / NW;                                // pieces p = wave + i NW < NPU
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_w3_zero_page);
    const uint16_t *xb = a.x + (int64_t)b * a.Cin * HW, *dyb = a.dy + (int64_t)b * a.Cout * HW;

    // Everything a fragment read can touch must be FINITE before the first MFMA (0 x NaN): slots not loaded yet, the slack
    // behind a row image, and - for even W / 8 - the RIGHT pad of a row, which is the left pad of the next channel / slot.
    for (int i = tid * 16; i < a.lds_bytes; i += NT * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);

    // ---- this wave's pieces of a unit: piece p = wave + i NW; x pieces first (row-major), then dY pieces.  Pieces 0 .. NPX - 1 are the
    // unit's new rows of X (r0 + 1 .. r0 + WK, r0 = ra + u WK: rows r0 - 1, r0 are in the ring already), pieces NPX .. NPU - 1 its
    // rows of dY (r0 .. r0 + WK - 1).  A full unit is ALWAYS np_mine pieces per wave (rows outside the image come from the page
    // of zeros), which is what the counted waits rely on.
    int koff[kMaxP];                              // element offset of the lane's chunk inside the image (row 0), -1: zeros
#pragma unroll
    for (int i = 0; i < kMaxP; ++i) {
        const int p = wave + i * NW;
        koff[i] = -1;
        if (p < NPU) {
            const bool isx = p < NPX;
            const int pp = isx ? p : p - NPX, ppr = isx ? PPRX : PPRD;
            const int pin = pp >= ppr ? pp - ppr : pp;
            const int q = pin * 64 + lane, ch = q / cpr, j = q - ch * cpr;
            const int cg = (isx ? c0 : n0) + ch;
            if (ch < (isx ? TC : TN) && cg < (isx ? a.Cin : a.Cout) && j >= 1 && j <= w8) koff[i] = cg * HW + (j - 1) * 8;
        }
    }
    auto issue_piece = [&](int u, int p, int ko) {           // ko = koff[] entry of piece p (the caller indexes it statically)
        const int r0 = ra + u * WK;
        const bool isx = p < NPX;
        const int pp = isx ? p : p - NPX, ppr = isx ? PPRX : PPRD;
        const int rowi = pp >= ppr ? 1 : 0, pin = pp - rowi * ppr;
        const int grow = isx ? r0 + 1 + rowi : r0 + rowi;
        const int slot = isx ? (grow + 1 - ra) % NXR : (grow - ra) % NDR;
        const bool ok = grow >= 0 && grow < H && ko >= 0;
        w3_glds16(ok ? (isx ? xb : dyb) + grow * W + ko : zero,
                  __builtin_amdgcn_readfirstlane(lds0 + (isx ? slot * XSLOT : XRING + slot * DSLOT) + pin * 1024));
    };
    auto issue_range = [&](int u, int i_lo, int i_hi, bool x_only) {     // pieces i_lo <= i < i_hi of this wave
#pragma unroll
        for (int i = 0; i < kMaxP; ++i) {
            const int p = wave + i * NW;
            if (i >= i_lo && i < i_hi && p < (x_only ? NPX : NPU)) issue_piece(u, p, koff[i]);
        }
    };

    w3_f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nunits = (rb - ra + WK - 1) / WK;
    __syncthreads();                                         // the zero fill is done before the first LDS-DMA may land
    // prologue: rows ra - 1, ra of X (the X pieces of the units "before" the first one), then PF whole units
    if (WK == 1) issue_range(-2, 0, kMaxP, true);
    issue_range(-1, 0, kMaxP, true);
#pragma unroll
    for (int v = 0; v < PF; ++v)
        if (v < nunits) issue_range(v, 0, kMaxP, false);
    const int h = lane >> 5, l31 = lane & 31;
    const int dlane = (tnw * 32 + l31) * CS + h * 16, xlane = (tcw * 32 + l31) * CS + (h + 1) * 16;
    const int nkr = (W + 15) >> 4;
    for (int u = 0; u < nunits; ++u) {
        // unit u's rows must have landed; the units behind it (u + 1 .. u + PF - 1, issued later, returned in order) may stay in flight
        w3_wait_vmcnt(min(PF - 1, nunits - 1 - u) * np_mine);
        __builtin_amdgcn_s_barrier();                        // ... every wave's pieces of it; and unit u - 1 is done with its slots
        const bool feed = u + PF < nunits && !(ablate & 1);  // unit u + PF goes into the slots unit u - 1 has released
        const int uf = u + PF;
        // OPT & 4 (needs PF >= 2): the waves of odd groups issue the unit's pieces BEFORE their MFMAs, the waves of even groups
        // AFTER theirs.  A SIMD hosts one wave of each kind (waves w and w + NW / 2), and a wave that issues 6 - 7 LDS-DMA pieces sits
        // in the vector-memory issue queue for several hundred cycles (the CU's address path takes ~30 cycles per 1 KiB piece):
        // with every wave issuing right behind the barrier the load time simply ADDED to the MFMA time (55 = 42 + 13 us on
        // 128 -> 128 @ 80 x 80); this way one wave of a SIMD computes while the other one issues.
        const bool late = (OPT & 4) && PF >= 2 && !(grp & 1);
        if (feed && !late) issue_range(uf, 0, kMaxP, false);
        const int r = ra + u * WK + k;
        if (r < rb && !(ablate & 2)) {
            const unsigned char *dbase = lds + XRING + ((r - ra) % NDR) * DSLOT + dlane;
            const unsigned char *xbase0 = lds + ((r + 0 - ra) % NXR) * XSLOT + xlane;
            const unsigned char *xbase1 = lds + ((r + 1 - ra) % NXR) * XSLOT + xlane;
            const unsigned char *xbase2 = lds + ((r + 2 - ra) % NXR) * XSLOT + xlane;
            auto ldfrag = [&](int s, W3Frag &f) {
                const int so = s * 32;
                f.ce = *reinterpret_cast<const uint4 *>(dbase + so + 16);
                f.ri = *reinterpret_cast<const uint4 *>(dbase + so + 32);
                f.x0 = *reinterpret_cast<const uint4 *>(xbase0 + so);
                f.x1 = *reinterpret_cast<const uint4 *>(xbase1 + so);
                f.x2 = *reinterpret_cast<const uint4 *>(xbase2 + so);
            };
            auto mma = [&](const W3Frag &f, uint32_t left) {
                // dY shifted by +1 (tap kc = 0) and by -1 (kc = 2): pixels 1..8 and -1..6 of the chunk
                const uint32_t m0 = __builtin_amdgcn_alignbit(f.ce.y, f.ce.x, 16), m1 = __builtin_amdgcn_alignbit(f.ce.z, f.ce.y, 16),
                               m2 = __builtin_amdgcn_alignbit(f.ce.w, f.ce.z, 16);
                const uint4 d0 = make_uint4(m0, m1, m2, __builtin_amdgcn_alignbit(f.ri.x, f.ce.w, 16));
                const uint4 d2 = make_uint4(__builtin_amdgcn_alignbit(f.ce.x, left, 16), m0, m1, m2);
                const w3_bf16x8 a0 = __builtin_bit_cast(w3_bf16x8, d0), a1 = __builtin_bit_cast(w3_bf16x8, f.ce),
                                a2 = __builtin_bit_cast(w3_bf16x8, d2);
                const w3_bf16x8 x0 = __builtin_bit_cast(w3_bf16x8, f.x0), x1 = __builtin_bit_cast(w3_bf16x8, f.x1),
                                x2 = __builtin_bit_cast(w3_bf16x8, f.x2);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x0, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x1, acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, acc[4], 0, 0, 0);
                acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x1, acc[5], 0, 0, 0);
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x2, acc[6], 0, 0, 0);
                acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x2, acc[7], 0, 0, 0);
                acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x2, acc[8], 0, 0, 0);
            };
            uint32_t left = 0;
            if (WS == 1) left = *reinterpret_cast<const uint32_t *>(dbase + 12);
            for (int s = ks; s < nkr; s += WS) {
                W3Frag f;
                if (WS != 1) left = *reinterpret_cast<const uint32_t *>(dbase + s * 32 + 12);
                ldfrag(s, f);
                mma(f, left);
                left = f.ri.w;
            }
        }
        if (feed && late) issue_range(uf, 0, kMaxP, false);
    }
    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    // (1) the G groups of a wave tile add up: groups > 0 park their accumulators in LDS ([register][lane]: conflict-free), group 0
    //     adds them, taps in two halves (5 + 4) to fit the ring's footprint.
    const int wt = tnw * TCW + tcw;                                              // wave tile
    const int NP16 = a.NP16, CP16 = a.CP16;
    if (G > 1) {
        float *red = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t0 = half * 5, nt_ = half ? 4 : 5;
            __syncthreads();                                                     // ring (or the previous half) no longer read
            if (grp > 0) {
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (t < nt_)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            red[(((grp - 1) * (TNW * TCW) + wt) * 80 + t * 16 + r) * 64 + lane] = acc[t0 + t][r];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll 1
                for (int g = 1; g < G; ++g)
#pragma unroll
                    for (int t = 0; t < 5; ++t)
                        if (t < nt_)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                acc[t0 + t][r] += red[(((g - 1) * (TNW * TCW) + wt) * 80 + t * 16 + r) * 64 + lane];
            }
        }
    }
    if (!(OPT & 1)) {
        // partial sums straight from the accumulators: part[split][n][c][9]; lane: c = l31, n = (r & 3) + 8 (r >> 2) + 4 h
        if (grp > 0) return;
        const int c = c0 + tcw * 32 + l31;
        if (c < CP16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + tnw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n >= NP16) continue;
                float *dst = a.part + (((int64_t)split * NP16 + n) * CP16 + c) * 9;
#pragma unroll
                for (int t = 0; t < 9; ++t) dst[t] = acc[t][r];
            }
        }
        return;
    }
    // (2) group 0 writes its sums into an LDS image of the tile in the slab's layout, [32 n][TC c][9 taps] per portion of 32
    //     output channels, which ALL waves copy out with 16-byte stores.
    float *img = reinterpret_cast<float *>(lds);
    const int cl = tcw * 32 + l31;
    const int rowf4 = TC * 9 / 4;                                                 // float4 per n row of the image
    const int cvalid4 = min(TC, CP16 - c0) * 9 / 4;
    for (int pn = 0; pn < TNW; ++pn) {
        __syncthreads();                                      // reduction / previous portion's copy-out done with the LDS
        if (tnw == pn && grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (r & 3) + 8 * (r >> 2) + 4 * h;
                float *e = img + (nl * TC + cl) * 9;
#pragma unroll
                for (int t = 0; t < 9; ++t) e[t] = acc[t][r];
            }
        }
        __syncthreads();
        for (int i = tid; i < 32 * rowf4; i += NT) {
            const int nl = i / rowf4, w4 = i - nl * rowf4;
            const int n = n0 + pn * 32 + nl;
            if (n < NP16 && w4 < cvalid4)
                *reinterpret_cast<float4 *>(a.part + (((int64_t)split * NP16 + n) * CP16 + c0) * 9 + 4 * w4) =
                    *reinterpret_cast<const float4 *>(img + (nl * TC) * 9 + 4 * w4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side
struct W3Plan {
    int cfg;              // 0: 64 x 64 tiles, 2 rows per unit (8 waves); 1: 64 x 64, 1 row (4 waves); 2: 32 x 32, 4 step groups (4 waves)
    int pf;               // units in flight
    int cpr, spi, rps, nsplits, ntiles, nct;
    size_t lds;
};

static size_t w3_ring_bytes(int T, int WK, int PF, int cpr) {
    const size_t ppr = ((size_t)T * cpr + 63) / 64;                  // TN == TC in every configuration
    return ((WK + 2 + PF * WK) + (PF + 1) * WK) * ppr * 1024 + 1024;
}

static int w3_env(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// The row-streaming kernel takes W % 8 == 0 (16-byte chunks), W <= 160, H >= 1.  Returns false for anything else.
bool wgrad3_rows_plan(int B, int Cin, int Cout, int H, int W, W3Plan *p) {
    static const int env = w3_env("DFINE_WGRAD3_ROWS", 1);
    if (!env || W % 8 || W < 8 || W > 160 || H < 1 || B < 1) return false;
    const int w8 = W / 8;
    const int cpr = (w8 & 1) ? w8 + 2 : w8 + 1;        // odd; for even W / 8 the right pad is the next row image's left pad
    const bool small = Cin <= 32 && Cout <= 32;
    const int T = small ? 32 : 64;
    // the epilogue's LDS needs: the tile image (32 output channels x T x 9 floats) and the parked accumulators of the other groups
    const size_t kLds = 160 * 1024, kOut = small ? (size_t)3 * 80 * 256 : (size_t)4 * 80 * 256;
    int cfg, pf;
    if (small) {
        cfg = 2; pf = 1;                                             // 2 workgroups per CU instead of a deeper ring
        if (w3_ring_bytes(32, 1, 1, cpr) > kLds) return false;
    } else {
        cfg = 0; pf = 3;
        while (pf > 1 && w3_ring_bytes(64, 2, pf, cpr) > kLds) --pf;
        if (w3_ring_bytes(64, 2, pf, cpr) > kLds) {
            cfg = 1; pf = 2;
            while (pf > 1 && w3_ring_bytes(64, 1, pf, cpr) > kLds) --pf;
            if (w3_ring_bytes(64, 1, pf, cpr) > kLds) return false;
        }
    }
    const int wk = cfg == 0 ? 2 : 1, nw = cfg == 0 ? 8 : 4;
    size_t lds = w3_ring_bytes(T, wk, pf, cpr);
    if (lds < kOut) lds = kOut;
    const int nnt = (Cout + T - 1) / T, nct = (Cin + T - 1) / T, ntiles = nnt * nct;
    // pieces per wave and unit must fit the kernel's static table
    const int npu = wk * 2 * ((T * cpr + 63) / 64);
    if ((npu + nw - 1) / nw > (cfg == 0 ? 8 : cfg == 1 ? 12 : 6)) return false;
    // splits: whole row ranges of ONE image; enough workgroups to fill the chip (one 8-wave / two 4-wave workgroups per CU),
    // fp32 partial sums (written once, read once by the deferred reduction) below ~24 MB or half the bytes of the operands
    const int64_t bytes_per_split = (int64_t)((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * 36;
    int64_t budget = (int64_t)B * H * W * (Cin + Cout);
    if (budget < 24000000) budget = 24000000;
    int cap = (int)(budget / bytes_per_split);
    if (cap < B) cap = B;
    const int target = cfg == 0 ? 256 : 512;
    int spi = (target / ntiles + B - 1) / B;
    if (spi < 1) spi = 1;
    if (spi * B > cap) spi = cap / B;
    if (spi < 1) spi = 1;
    const int min_rows = 2 * wk;                                   // at least two units per split
    if (spi > (H + min_rows - 1) / min_rows) spi = (H + min_rows - 1) / min_rows;
    if (spi < 1) spi = 1;
    int rps = (H + spi - 1) / spi;
    rps = (rps + wk - 1) / wk * wk;                                // whole units
    spi = (H + rps - 1) / rps;
    p->cfg = cfg; p->pf = pf; p->cpr = cpr; p->spi = spi; p->rps = rps; p->nsplits = spi * B; p->ntiles = ntiles; p->nct = nct;
    p->lds = lds;
    return true;
}

int wgrad3_rows_splits(int B, int Cin, int Cout, int H, int W) {
    W3Plan p;
    return wgrad3_rows_plan(B, Cin, Cout, H, W, &p) ? p.nsplits : 0;
}

template <int TNW, int TCW, int WK, int WS, int PF, int MAXP, int OPT>
static int w3_launch(const W3Args &a, const W3Plan &p, hipStream_t st) {
    static bool attr_set = false;                     // once per instantiation: not a stream operation, keep it out of graph capture
    auto kern = conv_wgrad3_rows_kernel<TNW, TCW, WK, WS, PF, MAXP, OPT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_last_error(e); return DFINE_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.nsplits * p.ntiles), dim3(64 * TNW * TCW * WK * WS), p.lds, st, a);
    return check_launch();
}

// part: [nsplits][NP16][CP16][9] fp32 partial sums (every element of the padded tile range written).
int wgrad3_rows_launch(const void *x, const void *dy, float *part, int B, int Cin, int Cout, int H, int W, hipStream_t st) {
    W3Plan p;
    if (!wgrad3_rows_plan(B, Cin, Cout, H, W, &p)) return DFINE_E_BADARG;
    W3Args a;
    a.x = (const uint16_t *)x; a.dy = (const uint16_t *)dy; a.part = part;
    a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.cpr = p.cpr; a.rps = p.rps; a.spi = p.spi; a.nct = p.nct;
    static const int abl = w3_env("DFINE_W3_ABLATE", 0);             // timing experiments only (1: no loads after the prologue, 2: no MFMAs)
    a.ablate = abl;
    a.lds_bytes = (int)p.lds;
    a.NP16 = (Cout + 15) / 16 * 16; a.CP16 = (Cin + 15) / 16 * 16; a.ntiles = p.ntiles; a.nsplits = p.nsplits;
#define W3_GO(TNW, TCW, WK, WS, PF, MAXP) return w3_launch<TNW, TCW, WK, WS, PF, MAXP, 5>(a, p, st);
    if (p.cfg == 0) {
        if (p.pf == 3) W3_GO(2, 2, 2, 1, 3, 8)
        if (p.pf == 2) W3_GO(2, 2, 2, 1, 2, 8)
        W3_GO(2, 2, 2, 1, 1, 8)
    }
    if (p.cfg == 1) {
        if (p.pf == 2) W3_GO(2, 2, 1, 1, 2, 12)
        W3_GO(2, 2, 1, 1, 1, 12)
    }
    W3_GO(1, 1, 1, 4, 1, 6)
#undef W3_GO
}

}  // namespace dfine
