// A1/A2 - depthwise k x k convolution (groups == channels), NCHW, forward + both gradients.
//
// Reference call sites: HGNetv2 LightConvBNAct.conv2 (5x5, stride 1) and HG_Stage.downsample
// (3x3, stride 2) - src/d_fine/arch/hgnetv2.py:96-105,295-303; HybridEncoder SCDown.cv2 (3x3,
// stride 2) - src/d_fine/arch/hybrid_encoder.py:96-103.  ATen routes these to MIOpen, which on
// gfx950/bf16 falls back to naive_conv kernels (forward, data grad) and a per-image
// im2col + GEMM loop (weight grad): ~20 ms per D-FINE-m step for 0.3 % of the FLOPs.
//
// These are HBM/L2-bound stencils: one (image, channel) plane is independent, a plane row is
// contiguous.  Each block owns a strip of output rows of one plane, stages the needed input rows
// (+halo) in LDS as fp32 with coalesced loads and computes from LDS; weights are read as fp32
// master parameters (no bf16 weight copy), accumulation is fp32.
//   dwconv_fwd   y[b,c,oy,ox]  = sum_{ky,kx} w[c,ky,kx] * x[b,c,oy*s+ky-p, ox*s+kx-p]
//   dwconv_dgrad dx[b,c,iy,ix] = sum_{ky,kx} w[c,ky,kx] * dy[b,c,(iy+p-ky)/s,(ix+p-kx)/s]  (exact division only)
//   dwconv_wgrad dw[c,ky,kx]   = sum_{b,oy,ox} dy[b,c,oy,ox] * x[b,c,oy*s+ky-p, ox*s+kx-p]  (fp32 atomics per block)
#include "common.h"

namespace dfine {

constexpr int kDwThreads = 256;
constexpr int kMaxK = 7;

// ---- forward: block = (plane, strip of TR output rows) ---------------------------------------
template <typename T, int KT, int ST>
__global__ __launch_bounds__(kDwThreads) void dwconv_fwd_kernel(
    const T *__restrict__ x, const float *__restrict__ w, T *__restrict__ y, int C, int H, int W,
    int OH, int OW, int Krt, int Srt, int P, int TR) {
    const int K = KT ? KT : Krt, S = ST ? ST : Srt;       // compile-time when specialised
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x;           // b * C + c
    const int c = plane % C;
    const int oy0 = blockIdx.y * TR;
    const int rows_out = min(TR, OH - oy0);
    const int iy0 = oy0 * S - P;
    const int rows_in = (rows_out - 1) * S + K;
    const int WP = W + 2 * P;               // padded row length in LDS
    float *tile = smem;                      // [rows_in][WP]
    float *wk = smem + rows_in * WP;         // [K*K]  (rows_in <= (TR-1)*S+K by construction)
    const T *xp = x + (int64_t)plane * H * W;
    for (int i = threadIdx.x; i < rows_in * WP; i += kDwThreads) {
        const int r = i / WP, cx = i - r * WP;
        const int iy = iy0 + r, ix = cx - P;
        tile[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? load_f(xp + (int64_t)iy * W + ix) : 0.f;
    }
    if (threadIdx.x < K * K) wk[threadIdx.x] = w[c * K * K + threadIdx.x];
    __syncthreads();
    T *yp = y + (int64_t)plane * OH * OW + (int64_t)oy0 * OW;
    for (int o = threadIdx.x; o < rows_out * OW; o += kDwThreads) {
        const int r = o / OW, ox = o - r * OW;
        const float *src = tile + (r * S) * WP + ox * S;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc += wk[ky * K + kx] * src[ky * WP + kx];
        store_f(yp + o, acc);
    }
}

// ---- data gradient: block = (plane, strip of TR input rows) ----------------------------------
template <typename T, int KT, int ST>
__global__ __launch_bounds__(kDwThreads) void dwconv_dgrad_kernel(
    const T *__restrict__ dy, const float *__restrict__ w, T *__restrict__ dx, int C, int H, int W,
    int OH, int OW, int Krt, int Srt, int P, int TR) {
    const int K = KT ? KT : Krt, S = ST ? ST : Srt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x;
    const int c = plane % C;
    const int iy0 = blockIdx.y * TR;
    const int rows_in = min(TR, H - iy0);
    // output rows that can touch input rows [iy0, iy0+rows_in): oy in [ceil((iy0+P-K+1)/S), floor((iy0+rows_in-1+P)/S)]
    int oy_lo = iy0 + P - K + 1;
    oy_lo = oy_lo <= 0 ? 0 : (oy_lo + S - 1) / S;
    int oy_hi = (iy0 + rows_in - 1 + P) / S;
    oy_hi = min(oy_hi, OH - 1);
    const int rows_dy = max(oy_hi - oy_lo + 1, 0);
    float *tile = smem;                       // [rows_dy][OW]
    float *wk = smem + rows_dy * OW;
    const T *dyp = dy + (int64_t)plane * OH * OW + (int64_t)oy_lo * OW;
    for (int i = threadIdx.x; i < rows_dy * OW; i += kDwThreads) tile[i] = load_f(dyp + i);
    if (threadIdx.x < K * K) wk[threadIdx.x] = w[c * K * K + threadIdx.x];
    __syncthreads();
    T *dxp = dx + (int64_t)plane * H * W + (int64_t)iy0 * W;
    for (int o = threadIdx.x; o < rows_in * W; o += kDwThreads) {
        const int r = o / W, ix = o - r * W;
        const int iy = iy0 + r;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ty = iy + P - ky;
            const int oy = ty / S;
            const bool vy = ty >= 0 && ty - oy * S == 0 && oy >= oy_lo && oy <= oy_hi;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int tx = ix + P - kx;
                const int ox = tx / S;
                const bool v = vy && tx >= 0 && tx - ox * S == 0 && ox < OW;
                const float t = tile[v ? (oy - oy_lo) * OW + ox : 0];
                acc += v ? wk[ky * K + kx] * t : 0.f;
            }
        }
        store_f(dxp + o, acc);
    }
}

// ---- weight gradient: block = (channel, chunk of images); K*K partial sums per thread ---------
template <typename T, int KK>
__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_kernel(
    const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ dw, int B, int C, int H,
    int W, int OH, int OW, int S, int P, int imgs_per_block) {
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * imgs_per_block, b1 = min(B, b0 + imgs_per_block);
    float acc[KK * KK];
#pragma unroll
    for (int i = 0; i < KK * KK; ++i) acc[i] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const T *xp = x + ((int64_t)b * C + c) * H * W;
        const T *dyp = dy + ((int64_t)b * C + c) * OH * OW;
        for (int o = threadIdx.x; o < OH * OW; o += kDwThreads) {
            const int oy = o / OW, ox = o - oy * OW;
            const float g = load_f(dyp + o);
            const int iy = oy * S - P, ix = ox * S - P;
#pragma unroll
            for (int ky = 0; ky < KK; ++ky) {
                const int yy = iy + ky;
                const bool vy = yy >= 0 && yy < H;
#pragma unroll
                for (int kx = 0; kx < KK; ++kx) {
                    const int xx = ix + kx;
                    const float v = (vy && xx >= 0 && xx < W) ? load_f(xp + (int64_t)yy * W + xx) : 0.f;
                    acc[ky * KK + kx] += g * v;
                }
            }
        }
    }
    __shared__ float red[kDwThreads / 64][KK * KK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < KK * KK; ++i) {
        float v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < KK * KK) {
        float v = 0.f;
        for (int wv = 0; wv < kDwThreads / 64; ++wv) v += red[wv][threadIdx.x];
        unsafeAtomicAdd(dw + c * KK * KK + threadIdx.x, v);
    }
}

// ---- weight gradient, LDS-tiled: the x plane (+halo) and the dy plane of one (image, channel) are
// staged once, then every thread accumulates its K*K taps from LDS (the direct kernel above re-reads
// each x element K*K/S^2 times from L1/L2).
template <typename T, int KK, int SS>
__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_lds_kernel(
    const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ dw, int B, int C, int H,
    int W, int OH, int OW, int P, int imgs_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * imgs_per_block, b1 = min(B, b0 + imgs_per_block);
    const int HP = H + 2 * P, WP = W + 2 * P;
    float *xt = smem;                  // [HP][WP], zero border
    float *gt = smem + HP * WP;        // [OH][OW]
    for (int i = threadIdx.x; i < HP * WP; i += kDwThreads) xt[i] = 0.f;
    float acc[KK * KK];
#pragma unroll
    for (int i = 0; i < KK * KK; ++i) acc[i] = 0.f;
    __syncthreads();
    for (int b = b0; b < b1; ++b) {
        const T *xp = x + ((int64_t)b * C + c) * H * W;
        const T *dyp = dy + ((int64_t)b * C + c) * OH * OW;
        for (int i = threadIdx.x; i < H * W; i += kDwThreads) {
            const int yy = i / W, xx = i - yy * W;
            xt[(yy + P) * WP + xx + P] = load_f(xp + i);
        }
        for (int i = threadIdx.x; i < OH * OW; i += kDwThreads) gt[i] = load_f(dyp + i);
        __syncthreads();
        for (int o = threadIdx.x; o < OH * OW; o += kDwThreads) {
            const int oy = o / OW, ox = o - oy * OW;
            const float g = gt[o];
            const float *src = xt + (oy * SS) * WP + ox * SS;
#pragma unroll
            for (int ky = 0; ky < KK; ++ky)
#pragma unroll
                for (int kx = 0; kx < KK; ++kx) acc[ky * KK + kx] += g * src[ky * WP + kx];
        }
        __syncthreads();
    }
    float *red = smem;                 // reuse
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < KK * KK; ++i) {
        float v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave * KK * KK + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < KK * KK) {
        float v = 0.f;
        for (int wv = 0; wv < kDwThreads / 64; ++wv) v += red[wv * KK * KK + threadIdx.x];
        unsafeAtomicAdd(dw + c * KK * KK + threadIdx.x, v);
    }
}

// ---- stride-1 "same" depthwise conv, bf16, register-tiled: a thread owns VW consecutive outputs of a row --------
// The generic kernels above read K*K LDS words (+ K*K weights) per output: LDS-bound at ~0.7 TB/s effective on the
// 5x5 layers of HGNetv2 stages 3/4 (128 ch @ 40x40, 256 ch @ 20x20, 16 layers per step).  Here the padded input
// rows sit in LDS as fp32 with the data starting at column 4, so the VW + K - 1 inputs a thread needs from a row
// are VW/4 + 2 aligned 16-byte reads, the weights live in registers (block-uniform), and global traffic is
// 16-byte (VW = 8) or 8-byte (VW = 4) vectors.  FLIP = true reverses the taps: the data gradient.
template <int VW> struct DwVec;
template <> struct DwVec<8> {
    typedef uint4 raw;
    static __device__ __forceinline__ void unpack(const uint4 &v, float (&o)[8]) {
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint4 pack(const float (&o)[8]) {
        uint4 r;
        r.x = pack_bf16x2(o[0], o[1]);
        r.y = pack_bf16x2(o[2], o[3]);
        r.z = pack_bf16x2(o[4], o[5]);
        r.w = pack_bf16x2(o[6], o[7]);
        return r;
    }
};
template <> struct DwVec<4> {
    typedef uint2 raw;
    static __device__ __forceinline__ void unpack(const uint2 &v, float (&o)[4]) {
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    static __device__ __forceinline__ uint2 pack(const float (&o)[4]) {
        uint2 r;
        r.x = pack_bf16x2(o[0], o[1]);
        r.y = pack_bf16x2(o[2], o[3]);
        return r;
    }
};

// stage rows [y0 - P, y0 - P + rows_in) of one plane into tile[rows_in][W + 8] (fp32, data at column 4, zero borders)
template <int VW>
__device__ __forceinline__ void dw_stage_rows(const uint16_t *plane, float *tile, int y_first, int rows_in, int H, int W) {
    const int WPD = W + 8, nv = W / VW;
    for (int i = threadIdx.x; i < rows_in * 2; i += kDwThreads) {      // left / right zero borders
        const int r = i >> 1;
        float4 *z = reinterpret_cast<float4 *>(tile + r * WPD + ((i & 1) ? W + 4 : 0));
        *z = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = threadIdx.x; i < rows_in * nv; i += kDwThreads) {
        const int r = i / nv, v = i - r * nv;
        const int gy = y_first + r;
        float o[VW];
        if (gy >= 0 && gy < H) {
            const typename DwVec<VW>::raw raw = *reinterpret_cast<const typename DwVec<VW>::raw *>(plane + (int64_t)gy * W + v * VW);
            DwVec<VW>::unpack(raw, o);
        } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = 0.f;
        }
        float4 *d = reinterpret_cast<float4 *>(tile + r * WPD + 4 + v * VW);
#pragma unroll
        for (int q = 0; q < VW / 4; ++q) d[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
}

template <int K, int VW, bool FLIP>
__global__ __launch_bounds__(kDwThreads) void dwconv_s1_vec_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                                  uint16_t *__restrict__ y, int C, int H, int W, int TR) {
    constexpr int P = K / 2, NQ = VW / 4 + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x, c = plane % C;
    const int y0 = blockIdx.y * TR;
    const int rows_out = min(TR, H - y0), rows_in = rows_out + K - 1;
    const int WPD = W + 8, nv = W / VW;
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = w[c * K * K + (FLIP ? K * K - 1 - i : i)];      // block-uniform
    const uint16_t *xp = x + (int64_t)plane * H * W;
    dw_stage_rows<VW>(xp, smem, y0 - P, rows_in, H, W);
    __syncthreads();
    uint16_t *yp = y + (int64_t)plane * H * W;
    for (int s = threadIdx.x; s < rows_out * nv; s += kDwThreads) {
        const int r = s / nv, v = s - r * nv;
        float acc[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const float4 *src = reinterpret_cast<const float4 *>(smem + (r + ky) * WPD + v * VW);
            float in[4 * NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const float4 t = src[q]; in[4 * q] = t.x; in[4 * q + 1] = t.y; in[4 * q + 2] = t.z; in[4 * q + 3] = t.w; }
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int e = 0; e < VW; ++e) acc[e] = fmaf(wk[ky * K + kx], in[e + kx + 4 - P], acc[e]);
        }
        *reinterpret_cast<typename DwVec<VW>::raw *>(yp + (int64_t)(y0 + r) * W + v * VW) = DwVec<VW>::pack(acc);
    }
}

// weight gradient of the same layers: block = (channel, chunk of images); the whole padded plane of x in LDS, the
// thread's VW dy values straight from global; K*K accumulators per thread, block reduce, atomics over image chunks.
template <int K, int VW>
__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_s1_vec_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                                        float *__restrict__ dw, int B, int C, int H, int W,
                                                                        int imgs_per_block) {
    constexpr int P = K / 2, NQ = VW / 4 + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * imgs_per_block, b1 = min(B, b0 + imgs_per_block);
    const int WPD = W + 8, nv = W / VW, rows_in = H + K - 1;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int64_t pl = ((int64_t)b * C + c) * H * W;
        dw_stage_rows<VW>(x + pl, smem, -P, rows_in, H, W);
        __syncthreads();
        for (int s = threadIdx.x; s < H * nv; s += kDwThreads) {
            const int r = s / nv, v = s - r * nv;
            float g[VW];
            DwVec<VW>::unpack(*reinterpret_cast<const typename DwVec<VW>::raw *>(dy + pl + (int64_t)r * W + v * VW), g);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const float4 *src = reinterpret_cast<const float4 *>(smem + (r + ky) * WPD + v * VW);
                float in[4 * NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) { const float4 t = src[q]; in[4 * q] = t.x; in[4 * q + 1] = t.y; in[4 * q + 2] = t.z; in[4 * q + 3] = t.w; }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    float a = acc[ky * K + kx];
#pragma unroll
                    for (int e = 0; e < VW; ++e) a = fmaf(g[e], in[e + kx + 4 - P], a);
                    acc[ky * K + kx] = a;
                }
            }
        }
        __syncthreads();
    }
    float *red = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        float v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave * K * K + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < K * K) {
        float v = 0.f;
        for (int wv = 0; wv < kDwThreads / 64; ++wv) v += red[wv * K * K + threadIdx.x];
        unsafeAtomicAdd(dw + c * K * K + threadIdx.x, v);
    }
}

// ---- 3x3 / stride 2 / pad 1 depthwise (HG_Stage.downsample, SCDown.cv2), bf16, register-tiled -------------------
// forward: a thread owns 4 consecutive outputs (inputs 2x-1 .. 2x+7 of three rows = 3 aligned 16-byte LDS reads per row)
__global__ __launch_bounds__(kDwThreads) void dwconv_s2_fwd_vec_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                                      uint16_t *__restrict__ y, int C, int H, int W, int TRo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x, c = plane % C;
    const int OH = H / 2, OW = W / 2;
    const int oy0 = blockIdx.y * TRo;
    const int rows_out = min(TRo, OH - oy0), rows_in = 2 * rows_out + 1;
    const int WPD = W + 8, nv = OW / 4;
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    dw_stage_rows<8>(x + (int64_t)plane * H * W, smem, 2 * oy0 - 1, rows_in, H, W);
    __syncthreads();
    uint16_t *yp = y + (int64_t)plane * OH * OW;
    for (int s = threadIdx.x; s < rows_out * nv; s += kDwThreads) {
        const int r = s / nv, v = s - r * nv;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float4 *src = reinterpret_cast<const float4 *>(smem + (2 * r + ky) * WPD + v * 8);
            float in[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const float4 t = src[q]; in[4 * q] = t.x; in[4 * q + 1] = t.y; in[4 * q + 2] = t.z; in[4 * q + 3] = t.w; }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk[ky * 3 + kx], in[2 * e + kx + 3], acc[e]);
        }
        *reinterpret_cast<uint2 *>(yp + (int64_t)(oy0 + r) * OW + v * 4) = DwVec<4>::pack(acc);
    }
}

// data gradient: a thread owns 8 consecutive dx of one row; row parity picks the kernel rows, column parity the columns
template <int VD>          // vector width used to stage dy rows (OW % VD == 0)
__global__ __launch_bounds__(kDwThreads) void dwconv_s2_dgrad_vec_kernel(const uint16_t *__restrict__ dy, const float *__restrict__ w,
                                                                        uint16_t *__restrict__ dx, int C, int H, int W, int TR) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x, c = plane % C;
    const int OH = H / 2, OW = W / 2;
    const int y0 = blockIdx.y * TR;                              // TR even
    const int rows = min(TR, H - y0);
    const int oy_lo = y0 / 2, rows_dy = rows / 2 + 1;
    const int WPD = OW + 8, nv = W / 8;
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    dw_stage_rows<VD>(dy + (int64_t)plane * OH * OW, smem, oy_lo, rows_dy, OH, OW);    // rows >= OH read as zeros
    __syncthreads();
    uint16_t *dxp = dx + (int64_t)plane * H * W;
    for (int s = threadIdx.x; s < rows * nv; s += kDwThreads) {
        const int r = s / nv, v = s - r * nv;
        const int yy = y0 + r;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // yy even: ky = 1, oy = yy / 2.   yy odd: ky = 0 -> oy = (yy + 1) / 2, ky = 2 -> oy = (yy - 1) / 2
        const int nky = (yy & 1) ? 2 : 1;
        for (int t = 0; t < nky; ++t) {
            const int ky = (yy & 1) ? 2 * t : 1;
            const int oy = (yy + 1 - ky) >> 1;
            const float4 *src = reinterpret_cast<const float4 *>(smem + (oy - oy_lo) * WPD + 4 + v * 4);
            const float4 a = src[0], b = src[1];
            const float g[5] = {a.x, a.y, a.z, a.w, b.x};       // dy columns c0 .. c0 + 4 (c0 = 4 v)
            const float w0 = wk[ky * 3], w1 = wk[ky * 3 + 1], w2 = wk[ky * 3 + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] = fmaf(w1, g[j], acc[2 * j]);                                   // x = 2c     : kx = 1
                acc[2 * j + 1] = fmaf(w0, g[j + 1], fmaf(w2, g[j], acc[2 * j + 1]));       // x = 2c + 1 : kx = 0 (c+1), kx = 2 (c)
            }
        }
        *reinterpret_cast<uint4 *>(dxp + (int64_t)yy * W + v * 8) = DwVec<8>::pack(acc);
    }
}

// weight gradient: block = (channel, image chunk); padded x plane in LDS, 4 dy values per thread from global
__global__ __launch_bounds__(kDwThreads) void dwconv_s2_wgrad_vec_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                                        float *__restrict__ dw, int B, int C, int H, int W,
                                                                        int imgs_per_block, int TRo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * imgs_per_block, b1 = min(B, b0 + imgs_per_block);
    const int OH = H / 2, OW = W / 2, WPD = W + 8, nv = OW / 4;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int64_t pl = (int64_t)b * C + c;
        for (int oy0 = 0; oy0 < OH; oy0 += TRo) {                 // row tiles: the whole plane need not fit in LDS
            const int rows_out = min(TRo, OH - oy0);
            dw_stage_rows<8>(x + pl * H * W, smem, 2 * oy0 - 1, 2 * rows_out + 1, H, W);
            __syncthreads();
            for (int s = threadIdx.x; s < rows_out * nv; s += kDwThreads) {
                const int r = s / nv, v = s - r * nv;
                float g[4];
                DwVec<4>::unpack(*reinterpret_cast<const uint2 *>(dy + pl * OH * OW + (int64_t)(oy0 + r) * OW + v * 4), g);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float4 *src = reinterpret_cast<const float4 *>(smem + (2 * r + ky) * WPD + v * 8);
                    float in[12];
#pragma unroll
                    for (int q = 0; q < 3; ++q) { const float4 t = src[q]; in[4 * q] = t.x; in[4 * q + 1] = t.y; in[4 * q + 2] = t.z; in[4 * q + 3] = t.w; }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float a = acc[ky * 3 + kx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a = fmaf(g[e], in[2 * e + kx + 3], a);
                        acc[ky * 3 + kx] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    float *red = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        float v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave * 9 + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        float v = 0.f;
        for (int wv = 0; wv < kDwThreads / 64; ++wv) v += red[wv * 9 + threadIdx.x];
        unsafeAtomicAdd(dw + c * 9 + threadIdx.x, v);
    }
}

// ---- 3x3 / stride 2 / pad 1, bf16, STREAMING form (no LDS) -----------------------------------------------------------------
// The strip kernels above move 1.7 - 2.9 TB/s stand-alone (tools/dwconv_bench.py): every block loads one small tile, waits, computes,
// stores - one tile's latency per block and ~10 KB per block of traffic.  Here a lane owns one 16-byte column vector of a plane
// (8 input columns = 4 outputs) and walks DOWN the rows with the previous input row kept in registers, a wave holds 64 / (W / 8)
// whole planes side by side, the column halo comes from the neighbouring lane (one DPP-style shuffle per row), and the loads of
// several output rows are issued before the first is consumed: a stream with constant bytes in flight instead of load-wait-compute
// tiles.  Rows are cut into chunks (blockIdx.y) for parallelism: one halo row re-read per chunk.
constexpr int kDwRowsAhead = 4;

// eval-mode BatchNorm (folded to a per-channel scale / shift) + activation + learnable affine on the fp32 sums before the store:
// y = lab[0] * act(scale[c] * conv + shift[c]) + lab[1] - the inference form of LightConvBNAct's depthwise unit as one launch
// (ref hgnetv2.py:83-112).  scale == nullptr: plain store.
struct DwEpi {
    const float *scale, *shift, *lab;
    int act;
};
__device__ __forceinline__ float dw_epi_act(float z, int act) {
    if (act == 1) return fmaxf(z, 0.f);
    if (act == 2) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
    return z;
}

__device__ __forceinline__ void dw_unpack9(const uint4 &r, uint32_t left_pair, bool first, float (&o)[9]) {
    // o[0] = column 8 v - 1 (the last element of the left neighbour's vector, zero at the plane edge), o[1..8] = own 8 columns
    o[0] = first ? 0.f : __uint_as_float(left_pair & 0xffff0000u);
    o[1] = __uint_as_float(r.x << 16); o[2] = __uint_as_float(r.x & 0xffff0000u);
    o[3] = __uint_as_float(r.y << 16); o[4] = __uint_as_float(r.y & 0xffff0000u);
    o[5] = __uint_as_float(r.z << 16); o[6] = __uint_as_float(r.z & 0xffff0000u);
    o[7] = __uint_as_float(r.w << 16); o[8] = __uint_as_float(r.w & 0xffff0000u);
}

// forward: y[oy][4 v + e] = sum_{ky, kx} w[ky][kx] x[2 oy + ky - 1][8 v + 2 e + kx - 1]
template <bool EPI>          // EPI: the inference epilogue (a separate instantiation: the training kernel keeps its 69 registers)
__global__ __launch_bounds__(kDwThreads) void dwconv_s2_fwd_stream_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                                         uint16_t *__restrict__ y, int C, int H, int W, int planes,
                                                                         int rows_per_chunk, const DwEpi ep) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (kDwThreads / 64) + (threadIdx.x >> 6);
    const int nv = W / 8, ppw = 64 / nv;
    const int pl = lane / nv, v = lane - pl * nv;
    const int plane = wv * ppw + pl;
    const bool live = pl < ppw && plane < planes;
    const int OH = H / 2, OW = W / 2;
    const int oy0 = blockIdx.y * rows_per_chunk, oy1 = min(OH, oy0 + rows_per_chunk);
    const int c = live ? plane % C : 0;
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    constexpr bool epi = EPI;
    const float e_sc = epi ? ep.scale[c] : 1.f, e_sh = epi ? ep.shift[c] : 0.f;
    const float e_ls = (epi && ep.lab) ? ep.lab[0] : 1.f, e_lb = (epi && ep.lab) ? ep.lab[1] : 0.f;
    const uint16_t *xp = x + (int64_t)(live ? plane : 0) * H * W + v * 8;
    uint16_t *yp = y + (int64_t)(live ? plane : 0) * OH * OW + v * 4;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto ld = [&](int row) { return (live && row >= 0) ? *reinterpret_cast<const uint4 *>(xp + (int64_t)row * W) : zero4; };
    uint4 rp = ld(2 * oy0 - 1);
    for (int oy = oy0; oy < oy1; oy += kDwRowsAhead) {
        uint4 ra[kDwRowsAhead], rb[kDwRowsAhead];
#pragma unroll
        for (int u = 0; u < kDwRowsAhead; ++u) {
            const bool ok = oy + u < oy1;
            ra[u] = ok ? ld(2 * (oy + u)) : zero4;
            rb[u] = ok ? ld(2 * (oy + u) + 1) : zero4;
        }
#pragma unroll
        for (int u = 0; u < kDwRowsAhead; ++u) {
            if (oy + u >= oy1) break;                              // uniform
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const uint4 rows[3] = {rp, ra[u], rb[u]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const uint32_t lp = __shfl_up(rows[ky].w, 1, 64);
                float in[9];
                dw_unpack9(rows[ky], lp, v == 0, in);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk[ky * 3 + kx], in[2 * e + kx], acc[e]);
            }
            if (epi) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = e_ls * dw_epi_act(acc[e] * e_sc + e_sh, ep.act) + e_lb;
            }
            if (live) *reinterpret_cast<uint2 *>(yp + (int64_t)(oy + u) * OW) = DwVec<4>::pack(acc);
            rp = rb[u];
        }
    }
}

// data gradient: a lane owns 4 dy columns = 8 dx columns; dy row oy feeds dx rows 2 oy (ky = 1), 2 oy - 1 (ky = 0) and 2 oy + 1 (ky = 2)
// ACC: dx += (dx holds the gradient of the map's other consumer; this term is rounded to bf16 first, so the sum is the one
// autograd's add of the two bf16 maps would give)
template <bool ACC>
__global__ __launch_bounds__(kDwThreads) void dwconv_s2_dgrad_stream_kernel(const uint16_t *__restrict__ dy, const float *__restrict__ w,
                                                                           uint16_t *__restrict__ dx, int C, int H, int W, int planes,
                                                                           int rows_per_chunk) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (kDwThreads / 64) + (threadIdx.x >> 6);
    const int nv = W / 8, ppw = 64 / nv;
    const int pl = lane / nv, v = lane - pl * nv;
    const int plane = wv * ppw + pl;
    const bool live = pl < ppw && plane < planes;
    const int OH = H / 2, OW = W / 2;
    const int oy0 = blockIdx.y * rows_per_chunk, oy1 = min(OH, oy0 + rows_per_chunk);
    const int c = live ? plane % C : 0;
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    const uint16_t *gp = dy + (int64_t)(live ? plane : 0) * OH * OW + v * 4;
    uint16_t *dxp = dx + (int64_t)(live ? plane : 0) * H * W + v * 8;
    const uint2 zero2 = make_uint2(0, 0);
    auto ld = [&](int row) { return (live && row < OH) ? *reinterpret_cast<const uint2 *>(gp + (int64_t)row * OW) : zero2; };
    auto unpack5 = [&](const uint2 &r, float (&g)[5]) {
        const uint32_t right = __shfl_down(r.x, 1, 64);           // the right neighbour's first dy column
        g[0] = __uint_as_float(r.x << 16); g[1] = __uint_as_float(r.x & 0xffff0000u);
        g[2] = __uint_as_float(r.y << 16); g[3] = __uint_as_float(r.y & 0xffff0000u);
        g[4] = (v == nv - 1) ? 0.f : __uint_as_float(right << 16);
    };
    auto row_terms = [&](const float (&g)[5], int ky, float (&acc)[8]) {
        const float w0 = wk[ky * 3], w1 = wk[ky * 3 + 1], w2 = wk[ky * 3 + 2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[2 * j] = fmaf(w1, g[j], acc[2 * j]);                                   // x = 2c     : kx = 1
            acc[2 * j + 1] = fmaf(w0, g[j + 1], fmaf(w2, g[j], acc[2 * j + 1]));       // x = 2c + 1 : kx = 0 (c + 1), kx = 2 (c)
        }
    };
    float gc[5], gn[5];
    { const uint2 r = ld(oy0); unpack5(r, gc); }
    for (int oy = oy0; oy < oy1; oy += kDwRowsAhead) {
        uint2 rn[kDwRowsAhead];
#pragma unroll
        for (int u = 0; u < kDwRowsAhead; ++u) rn[u] = (oy + u < oy1) ? ld(oy + u + 1) : zero2;     // (row OH reads as zeros)
#pragma unroll
        for (int u = 0; u < kDwRowsAhead; ++u) {
            if (oy + u >= oy1) break;                              // uniform
            unpack5(rn[u], gn);
            float even[8], odd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { even[e] = 0.f; odd[e] = 0.f; }
            row_terms(gc, 1, even);                                // dx row 2 oy
            row_terms(gc, 2, odd);                                 // dx row 2 oy + 1: ky = 2 from dy row oy ...
            row_terms(gn, 0, odd);                                 // ... and ky = 0 from dy row oy + 1
            if (live) {
                uint4 *pe = reinterpret_cast<uint4 *>(dxp + (int64_t)(2 * (oy + u)) * W);
                uint4 *po = reinterpret_cast<uint4 *>(dxp + (int64_t)(2 * (oy + u) + 1) * W);
                if (ACC) {
                    float olde[8], oldo[8], te[8], to[8];
                    DwVec<8>::unpack(*pe, olde);
                    DwVec<8>::unpack(*po, oldo);
                    DwVec<8>::unpack(DwVec<8>::pack(even), te);
                    DwVec<8>::unpack(DwVec<8>::pack(odd), to);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { even[e] = te[e] + olde[e]; odd[e] = to[e] + oldo[e]; }
                }
                *pe = DwVec<8>::pack(even);
                *po = DwVec<8>::pack(odd);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) gc[i] = gn[i];
        }
    }
}

// weight gradient: a wave owns 64 / (W / 8) channels, walks the planes of its image chunk row by row, 9 sums per lane; at the end
// the lanes of a channel are added through LDS and go to dw with one atomic per (channel, tap, block)
__global__ __launch_bounds__(kDwThreads) void dwconv_s2_wgrad_stream_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                                           float *__restrict__ dw, int B, int C, int H, int W,
                                                                           int imgs_per_block, int rows_per_chunk) {
    __shared__ float red[kDwThreads][9 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wv = blockIdx.x * (kDwThreads / 64) + wave;
    const int nv = W / 8, ppw = 64 / nv;
    const int pl = lane / nv, v = lane - pl * nv;
    const int c = wv * ppw + pl;
    const bool live = pl < ppw && c < C;
    const int OH = H / 2, OW = W / 2;
    const int b0 = blockIdx.y * imgs_per_block, b1 = min(B, b0 + imgs_per_block);
    const int oy0 = blockIdx.z * rows_per_chunk, oy1 = min(OH, oy0 + rows_per_chunk);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int64_t plane = (int64_t)b * C + (live ? c : 0);
        const uint16_t *xp = x + plane * H * W + v * 8;
        const uint16_t *gp = dy + plane * OH * OW + v * 4;
        auto ld = [&](int row) { return (live && row >= 0) ? *reinterpret_cast<const uint4 *>(xp + (int64_t)row * W) : zero4; };
        uint4 rp = ld(2 * oy0 - 1);
        for (int oy = oy0; oy < oy1; oy += kDwRowsAhead) {
            uint4 ra[kDwRowsAhead], rb[kDwRowsAhead];
            uint2 rg[kDwRowsAhead];
#pragma unroll
            for (int u = 0; u < kDwRowsAhead; ++u) {
                const bool ok = oy + u < oy1;
                ra[u] = ok ? ld(2 * (oy + u)) : zero4;
                rb[u] = ok ? ld(2 * (oy + u) + 1) : zero4;
                rg[u] = (ok && live) ? *reinterpret_cast<const uint2 *>(gp + (int64_t)(oy + u) * OW) : make_uint2(0, 0);
            }
#pragma unroll
            for (int u = 0; u < kDwRowsAhead; ++u) {
                if (oy + u >= oy1) break;                          // uniform
                float g[4];
                DwVec<4>::unpack(rg[u], g);
                const uint4 rows[3] = {rp, ra[u], rb[u]};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const uint32_t lp = __shfl_up(rows[ky].w, 1, 64);
                    float in[9];
                    dw_unpack9(rows[ky], lp, v == 0, in);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float a = acc[ky * 3 + kx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a = fmaf(g[e], in[2 * e + kx], a);
                        acc[ky * 3 + kx] = a;
                    }
                }
                rp = rb[u];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    // thread t < 4 waves x ppw channels x 9 taps: sum of the nv lanes of that channel
    const int per_wave = ppw * 9;
    for (int t = threadIdx.x; t < (kDwThreads / 64) * per_wave; t += kDwThreads) {
        const int w_ = t / per_wave, r_ = t - w_ * per_wave, p_ = r_ / 9, i_ = r_ - p_ * 9;
        const int cc = (blockIdx.x * (kDwThreads / 64) + w_) * ppw + p_;
        if (cc >= C) continue;
        float sum = 0.f;
        for (int l = 0; l < nv; ++l) sum += red[w_ * 64 + p_ * nv + l][i_];
        unsafeAtomicAdd(dw + cc * 9 + i_, sum);
    }
}

// ---- stride 1 "same" layers (LightConvBNAct 5x5; 3x3), bf16, STREAMING form ------------------------------------------------
// Same scheme: a lane owns one VW-column vector of a plane and walks down the rows with a K-row window of fp32 values in registers
// (each input row is converted once), the P halo columns each side come from the neighbouring lanes' edge words.
template <int VW> struct DwRaw;
template <> struct DwRaw<8> {
    typedef uint4 T;
    static __device__ __forceinline__ T zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ uint32_t first(const T &r) { return r.x; }
    static __device__ __forceinline__ uint32_t last(const T &r) { return r.w; }
};
template <> struct DwRaw<4> {
    typedef uint2 T;
    static __device__ __forceinline__ T zero() { return make_uint2(0, 0); }
    static __device__ __forceinline__ uint32_t first(const T &r) { return r.x; }
    static __device__ __forceinline__ uint32_t last(const T &r) { return r.y; }
};

// o[j] = column VW v - 2 + j, j = 0 .. VW + 3 (two halo columns each side, zero outside the plane)
template <int VW>
__device__ __forceinline__ void dw_unpack_halo(const typename DwRaw<VW>::T &r, bool first, bool last, float (&o)[VW + 4]) {
    const uint32_t lw = __shfl_up(DwRaw<VW>::last(r), 1, 64), rw = __shfl_down(DwRaw<VW>::first(r), 1, 64);
    o[0] = first ? 0.f : __uint_as_float(lw << 16);
    o[1] = first ? 0.f : __uint_as_float(lw & 0xffff0000u);
    float c[VW];
    DwVec<VW>::unpack(r, c);
#pragma unroll
    for (int e = 0; e < VW; ++e) o[2 + e] = c[e];
    o[VW + 2] = last ? 0.f : __uint_as_float(rw << 16);
    o[VW + 3] = last ? 0.f : __uint_as_float(rw & 0xffff0000u);
}

template <int K, int VW, bool FLIP, bool EPI = false>   // EPI: the inference epilogue (forward only; its own instantiation)
__global__ __launch_bounds__(kDwThreads) void dwconv_s1_stream_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                                     uint16_t *__restrict__ y, int C, int H, int W, int planes,
                                                                     int rows_per_chunk, const DwEpi ep) {
    constexpr int P = K / 2;
    typedef typename DwRaw<VW>::T Raw;
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (kDwThreads / 64) + (threadIdx.x >> 6);
    const int nv = W / VW, ppw = 64 / nv;
    const int pl = lane / nv, v = lane - pl * nv;
    const int plane = wv * ppw + pl;
    const bool live = pl < ppw && plane < planes;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(H, r0 + rows_per_chunk);
    const int c = live ? plane % C : 0;
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = w[c * K * K + (FLIP ? K * K - 1 - i : i)];
    constexpr bool epi = EPI && !FLIP;
    const float e_sc = epi ? ep.scale[c] : 1.f, e_sh = epi ? ep.shift[c] : 0.f;
    const float e_ls = (epi && ep.lab) ? ep.lab[0] : 1.f, e_lb = (epi && ep.lab) ? ep.lab[1] : 0.f;
    const uint16_t *xp = x + (int64_t)(live ? plane : 0) * H * W + v * VW;
    uint16_t *yp = y + (int64_t)(live ? plane : 0) * H * W + v * VW;
    auto ld = [&](int row) { return (live && row >= 0 && row < H) ? *reinterpret_cast<const Raw *>(xp + (int64_t)row * W) : DwRaw<VW>::zero(); };
    const bool first = v == 0, last = v == nv - 1;
    float win[K][VW + 4];                                          // rows r - P .. r + P of the current output row r
#pragma unroll
    for (int k = 0; k < K - 1; ++k) { const Raw t = ld(r0 - P + k); dw_unpack_halo<VW>(t, first, last, win[k + 1]); }
    for (int r = r0; r < r1; r += kDwRowsAhead) {
        Raw nx[kDwRowsAhead];
#pragma unroll
        for (int u = 0; u < kDwRowsAhead; ++u) nx[u] = (r + u < r1) ? ld(r + u + P) : DwRaw<VW>::zero();
#pragma unroll
        for (int u = 0; u < kDwRowsAhead; ++u) {
            if (r + u >= r1) break;                                // uniform
#pragma unroll
            for (int k = 0; k < K - 1; ++k)
#pragma unroll
                for (int j = 0; j < VW + 4; ++j) win[k][j] = win[k + 1][j];
            dw_unpack_halo<VW>(nx[u], first, last, win[K - 1]);
            float acc[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int e = 0; e < VW; ++e) acc[e] = fmaf(wk[ky * K + kx], win[ky][e + kx + 2 - P], acc[e]);
            if (epi) {
#pragma unroll
                for (int e = 0; e < VW; ++e) acc[e] = e_ls * dw_epi_act(acc[e] * e_sc + e_sh, ep.act) + e_lb;
            }
            if (live) *reinterpret_cast<Raw *>(yp + (int64_t)(r + u) * W) = DwVec<VW>::pack(acc);
        }
    }
}

template <int K, int VW>
__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_s1_stream_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                                           float *__restrict__ dw, int B, int C, int H, int W,
                                                                           int imgs_per_block, int rows_per_chunk) {
    constexpr int P = K / 2, KK = K * K;
    typedef typename DwRaw<VW>::T Raw;
    __shared__ float red[kDwThreads][KK + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wv = blockIdx.x * (kDwThreads / 64) + wave;
    const int nv = W / VW, ppw = 64 / nv;
    const int pl = lane / nv, v = lane - pl * nv;
    const int c = wv * ppw + pl;
    const bool live = pl < ppw && c < C;
    const int b0 = blockIdx.y * imgs_per_block, b1 = min(B, b0 + imgs_per_block);
    const int r0 = blockIdx.z * rows_per_chunk, r1 = min(H, r0 + rows_per_chunk);
    const bool first = v == 0, last = v == nv - 1;
    float acc[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) acc[i] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int64_t plane = (int64_t)b * C + (live ? c : 0);
        const uint16_t *xp = x + plane * H * W + v * VW;
        const uint16_t *gp = dy + plane * H * W + v * VW;
        auto ld = [&](int row) { return (live && row >= 0 && row < H) ? *reinterpret_cast<const Raw *>(xp + (int64_t)row * W) : DwRaw<VW>::zero(); };
        float win[K][VW + 4];
#pragma unroll
        for (int k = 0; k < K - 1; ++k) { const Raw t = ld(r0 - P + k); dw_unpack_halo<VW>(t, first, last, win[k + 1]); }
        for (int r = r0; r < r1; r += kDwRowsAhead) {
            Raw nx[kDwRowsAhead], ng[kDwRowsAhead];
#pragma unroll
            for (int u = 0; u < kDwRowsAhead; ++u) {
                const bool ok = r + u < r1;
                nx[u] = ok ? ld(r + u + P) : DwRaw<VW>::zero();
                ng[u] = (ok && live) ? *reinterpret_cast<const Raw *>(gp + (int64_t)(r + u) * W) : DwRaw<VW>::zero();
            }
#pragma unroll
            for (int u = 0; u < kDwRowsAhead; ++u) {
                if (r + u >= r1) break;                            // uniform
#pragma unroll
                for (int k = 0; k < K - 1; ++k)
#pragma unroll
                    for (int j = 0; j < VW + 4; ++j) win[k][j] = win[k + 1][j];
                dw_unpack_halo<VW>(nx[u], first, last, win[K - 1]);
                float g[VW];
                DwVec<VW>::unpack(ng[u], g);
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        float a = acc[ky * K + kx];
#pragma unroll
                        for (int e = 0; e < VW; ++e) a = fmaf(g[e], win[ky][e + kx + 2 - P], a);
                        acc[ky * K + kx] = a;
                    }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KK; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    const int per_wave = ppw * KK;
    for (int t = threadIdx.x; t < (kDwThreads / 64) * per_wave; t += kDwThreads) {
        const int w_ = t / per_wave, r_ = t - w_ * per_wave, p_ = r_ / KK, i_ = r_ - p_ * KK;
        const int cc = (blockIdx.x * (kDwThreads / 64) + w_) * ppw + p_;
        if (cc >= C) continue;
        float sum = 0.f;
        for (int l = 0; l < nv; ++l) sum += red[w_ * 64 + p_ * nv + l][i_];
        unsafeAtomicAdd(dw + cc * KK + i_, sum);
    }
}

static bool dw_s2_ok(int dtype, int H, int W, int K, int stride, int pad) {
    return dtype == DFINE_BF16 && stride == 2 && K == 3 && pad == 1 && H % 2 == 0 && W % 8 == 0 && W <= 320;
}

// stride-1 "same" layers the register-tiled kernels cover
static bool dw_vec_ok(int dtype, int H, int W, int K, int stride, int pad, int *vw) {
    if (dtype != DFINE_BF16 || stride != 1 || (K != 3 && K != 5) || pad != K / 2 || W > 160) return false;
    *vw = (W % 8 == 0) ? 8 : (W % 4 == 0 ? 4 : 0);
    return *vw != 0 && (size_t)(H + K - 1) * (W + 8) * 4 <= 60 * 1024;
}

static int pick_rows(int rows_total, int row_len_lds, int K, int S, bool fwd) {
    // strip height so that the LDS tile stays <= ~48 KiB and there are enough blocks
    int tr = 16;
    while (tr > 1) {
        const int rows_in = fwd ? (tr - 1) * S + K : tr / S + K;
        if ((size_t)rows_in * row_len_lds * 4 <= 48 * 1024) break;
        tr >>= 1;
    }
    return tr > rows_total ? rows_total : tr;
}

// output rows per chunk of the streaming kernels: whole multiples of the rows-ahead depth, enough chunks for ~2048+ waves
static int dw_stream_rows(int OH, int waves) {
    int rpc = OH;
    while (rpc > 2 * kDwRowsAhead && (int64_t)waves * ((OH + rpc - 1) / rpc) < 4096) rpc = (rpc + 1) / 2;
    return (rpc + kDwRowsAhead - 1) / kDwRowsAhead * kDwRowsAhead;
}

}  // namespace dfine

using namespace dfine;

extern "C" {

// One-shot request consumed by the next dfine_dwconv_fwd of the calling thread: y = lab[0] * act(scale[c] * conv + shift[c]) + lab[1]
// on the fp32 sums before the store (act 0 none / 1 ReLU / 2 SiLU; lab 2 floats or NULL; scale == NULL withdraws the request).
// Served by the bf16 streaming kernels (dfine_dwconv_affine_supported); a launch that cannot returns DFINE_E_BADARG.
static thread_local DwEpi g_dw_epi = {nullptr, nullptr, nullptr, 0};
int dfine_dwconv_affine_once(const float *scale, const float *shift, const float *lab, int act) {
    g_dw_epi = DwEpi{nullptr, nullptr, nullptr, 0};
    if (!scale) return DFINE_OK;
    if (!shift || act < 0 || act > 2) return DFINE_E_BADARG;
    g_dw_epi = DwEpi{scale, shift, lab, act};
    return DFINE_OK;
}

int dfine_dwconv_affine_supported(int dtype, int H, int W, int K, int stride, int pad) {
    int vw = 0;
    if (dw_vec_ok(dtype, H, W, K, stride, pad, &vw) && W / vw <= 64) return 1;
    if (dw_vec_ok(dtype, H, W, K, stride, pad, &vw)) return 0;
    return dw_s2_ok(dtype, H, W, K, stride, pad) ? 1 : 0;
}

int dfine_dwconv_fwd(const void *x, const float *w, void *y, int dtype, int B, int C, int H, int W,
                     int K, int stride, int pad, void *stream) {
    const DwEpi ep = g_dw_epi;
    g_dw_epi = DwEpi{nullptr, nullptr, nullptr, 0};
    if (ep.scale && !dfine_dwconv_affine_supported(dtype, H, W, K, stride, pad)) return DFINE_E_BADARG;
    if (B == 0 || C == 0) return DFINE_OK;
    if (!x || !w || !y || K < 1 || K > kMaxK || stride < 1 || pad < 0) return DFINE_E_BADARG;
    const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
    if (OH < 1 || OW < 1) return DFINE_E_BADARG;
    hipStream_t st0 = (hipStream_t)stream;
    int vw = 0;
    if (dw_vec_ok(dtype, H, W, K, stride, pad, &vw) && W / vw <= 64) {
        const int nv = W / vw, ppw = 64 / nv, planes = B * C;
        const int waves = (planes + ppw - 1) / ppw;
        const int rpc = dw_stream_rows(H, waves);
        const dim3 gs((waves + 3) / 4, (H + rpc - 1) / rpc);
#define DFINE_DWS(KK, VV) { if (ep.scale) hipLaunchKernelGGL((dwconv_s1_stream_kernel<KK, VV, false, true>), gs, dim3(kDwThreads), 0, st0, \
                                                            (const uint16_t *)x, w, (uint16_t *)y, C, H, W, planes, rpc, ep);      \
                            else hipLaunchKernelGGL((dwconv_s1_stream_kernel<KK, VV, false, false>), gs, dim3(kDwThreads), 0, st0, \
                                                    (const uint16_t *)x, w, (uint16_t *)y, C, H, W, planes, rpc, ep); }
        if (K == 5) { if (vw == 8) DFINE_DWS(5, 8) else DFINE_DWS(5, 4) }
        else { if (vw == 8) DFINE_DWS(3, 8) else DFINE_DWS(3, 4) }
#undef DFINE_DWS
        return check_launch();
    }
    if (dw_vec_ok(dtype, H, W, K, stride, pad, &vw)) {
        // rows per block: as many as 256 threads cover with one strip each
        int TRv = kDwThreads / (W / vw);
        if (TRv > H) TRv = H;
        const size_t smv = sizeof(float) * (size_t)(TRv + K - 1) * (W + 8);
        dim3 gridv(B * C, (H + TRv - 1) / TRv);
#define DFINE_DWV(KK, VV) hipLaunchKernelGGL((dwconv_s1_vec_kernel<KK, VV, false>), gridv, dim3(kDwThreads), smv, st0, \
                                             (const uint16_t *)x, w, (uint16_t *)y, C, H, W, TRv)
        if (K == 5) { if (vw == 8) DFINE_DWV(5, 8); else DFINE_DWV(5, 4); }
        else { if (vw == 8) DFINE_DWV(3, 8); else DFINE_DWV(3, 4); }
#undef DFINE_DWV
        return check_launch();
    }
    if (dw_s2_ok(dtype, H, W, K, stride, pad)) {
        const int nv = W / 8, ppw = 64 / nv, planes = B * C;
        const int waves = (planes + ppw - 1) / ppw;
        const int rpc = dw_stream_rows(OH, waves);
        if (ep.scale)
            hipLaunchKernelGGL(dwconv_s2_fwd_stream_kernel<true>, dim3((waves + 3) / 4, (OH + rpc - 1) / rpc), dim3(kDwThreads), 0, st0,
                               (const uint16_t *)x, w, (uint16_t *)y, C, H, W, planes, rpc, ep);
        else
            hipLaunchKernelGGL(dwconv_s2_fwd_stream_kernel<false>, dim3((waves + 3) / 4, (OH + rpc - 1) / rpc), dim3(kDwThreads), 0, st0,
                               (const uint16_t *)x, w, (uint16_t *)y, C, H, W, planes, rpc, ep);
        return check_launch();
    }
    if (dw_s2_ok(dtype, H, W, K, stride, pad)) {
        int TRo = kDwThreads / (OW / 4);
        if (TRo < 1) TRo = 1;
        if (TRo > OH) TRo = OH;
        const size_t smv = sizeof(float) * (size_t)(2 * TRo + 1) * (W + 8);
        hipLaunchKernelGGL(dwconv_s2_fwd_vec_kernel, dim3(B * C, (OH + TRo - 1) / TRo), dim3(kDwThreads), smv, st0,
                           (const uint16_t *)x, w, (uint16_t *)y, C, H, W, TRo);
        return check_launch();
    }
    const int WP = W + 2 * pad;
    const int TR = pick_rows(OH, WP, K, stride, true);
    const size_t sm = sizeof(float) * ((size_t)((TR - 1) * stride + K) * WP + K * K);
    dim3 grid(B * C, (OH + TR - 1) / TR);
    hipStream_t st = (hipStream_t)stream;
#define DFINE_DWF(TT, KK, SS)                                                                     \
    hipLaunchKernelGGL((dwconv_fwd_kernel<TT, KK, SS>), grid, dim3(kDwThreads), sm, st, (const TT *)x, w, (TT *)y, C, H, \
                       W, OH, OW, K, stride, pad, TR)
#define DFINE_DWF_KS(TT)                                                                          \
    { if (K == 3 && stride == 2) DFINE_DWF(TT, 3, 2); else if (K == 5 && stride == 1) DFINE_DWF(TT, 5, 1);      \
      else if (K == 3 && stride == 1) DFINE_DWF(TT, 3, 1); else DFINE_DWF(TT, 0, 0); }
    if (dtype == DFINE_F32) DFINE_DWF_KS(float)
    else if (dtype == DFINE_BF16) DFINE_DWF_KS(uint16_t)
    else return DFINE_E_BADARG;
#undef DFINE_DWF_KS
#undef DFINE_DWF
    return check_launch();
}

// dx += the data gradient of the 3x3 / stride-2 / pad-1 depthwise convolution (bf16; H even, W % 8 == 0, W <= 320: DFINE_E_BADARG
// otherwise - the caller then adds separately).  HG_Stage.downsample is the second consumer of the previous stage's output, which
// also leaves the backbone (ref hgnetv2.py:295-303,520-526): the encoder's gradient is already in dx.
int dfine_dwconv_s2_dgrad_acc(const float *w, const void *dy, void *dx, int B, int C, int H, int W, void *stream) {
    if (B == 0 || C == 0) return DFINE_OK;
    if (!w || !dy || !dx || !dw_s2_ok(DFINE_BF16, H, W, 3, 2, 1)) return DFINE_E_BADARG;
    const int OH = H / 2;
    const int nv = W / 8, ppw = 64 / nv, planes = B * C;
    const int waves = (planes + ppw - 1) / ppw;
    const int rpc = dw_stream_rows(OH, waves);
    hipLaunchKernelGGL(dwconv_s2_dgrad_stream_kernel<true>, dim3((waves + 3) / 4, (OH + rpc - 1) / rpc), dim3(kDwThreads), 0,
                       (hipStream_t)stream, (const uint16_t *)dy, w, (uint16_t *)dx, C, H, W, planes, rpc);
    return check_launch();
}

int dfine_dwconv_bwd(const void *x, const float *w, const void *dy, void *dx, float *dw_f32, int dtype,
                     int B, int C, int H, int W, int K, int stride, int pad, void *stream) {
    if (B == 0 || C == 0) return DFINE_OK;
    if (!w || !dy || K < 1 || K > kMaxK || stride < 1 || pad < 0) return DFINE_E_BADARG;
    if (dtype != DFINE_F32 && dtype != DFINE_BF16) return DFINE_E_BADARG;
    const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
    if (OH < 1 || OW < 1) return DFINE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int vw = 0;
    const bool vec = dw_vec_ok(dtype, H, W, K, stride, pad, &vw);
    if (dx && vec && W / vw <= 64) {
        const int nv = W / vw, ppw = 64 / nv, planes = B * C;
        const int waves = (planes + ppw - 1) / ppw;
        const int rpc = dw_stream_rows(H, waves);
        const dim3 gs((waves + 3) / 4, (H + rpc - 1) / rpc);
#define DFINE_DWS(KK, VV) hipLaunchKernelGGL((dwconv_s1_stream_kernel<KK, VV, true>), gs, dim3(kDwThreads), 0, st, \
                                             (const uint16_t *)dy, w, (uint16_t *)dx, C, H, W, planes, rpc, DwEpi{nullptr, nullptr, nullptr, 0})
        if (K == 5) { if (vw == 8) DFINE_DWS(5, 8); else DFINE_DWS(5, 4); }
        else { if (vw == 8) DFINE_DWS(3, 8); else DFINE_DWS(3, 4); }
#undef DFINE_DWS
        if (int e = check_launch()) return e;
    } else if (dx && vec) {
        int TRv = kDwThreads / (W / vw);
        if (TRv > H) TRv = H;
        const size_t smv = sizeof(float) * (size_t)(TRv + K - 1) * (W + 8);
        dim3 gridv(B * C, (H + TRv - 1) / TRv);
#define DFINE_DWV(KK, VV) hipLaunchKernelGGL((dwconv_s1_vec_kernel<KK, VV, true>), gridv, dim3(kDwThreads), smv, st, \
                                             (const uint16_t *)dy, w, (uint16_t *)dx, C, H, W, TRv)
        if (K == 5) { if (vw == 8) DFINE_DWV(5, 8); else DFINE_DWV(5, 4); }
        else { if (vw == 8) DFINE_DWV(3, 8); else DFINE_DWV(3, 4); }
#undef DFINE_DWV
        if (int e = check_launch()) return e;
    } else if (dx && dw_s2_ok(dtype, H, W, K, stride, pad)) {
        const int nv = W / 8, ppw = 64 / nv, planes = B * C;
        const int waves = (planes + ppw - 1) / ppw;
        const int rpc = dw_stream_rows(OH, waves);
        hipLaunchKernelGGL(dwconv_s2_dgrad_stream_kernel<false>, dim3((waves + 3) / 4, (OH + rpc - 1) / rpc), dim3(kDwThreads), 0, st,
                           (const uint16_t *)dy, w, (uint16_t *)dx, C, H, W, planes, rpc);
        if (int e = check_launch()) return e;
    } else if (dx && dw_s2_ok(dtype, H, W, K, stride, pad)) {
        int TRd = kDwThreads / (W / 8);
        TRd = TRd < 2 ? 2 : (TRd & ~1);
        if (TRd > H) TRd = H;
        const size_t smv = sizeof(float) * (size_t)(TRd / 2 + 1) * (OW + 8);
        dim3 gridv(B * C, (H + TRd - 1) / TRd);
        if (OW % 8 == 0)
            hipLaunchKernelGGL(dwconv_s2_dgrad_vec_kernel<8>, gridv, dim3(kDwThreads), smv, st, (const uint16_t *)dy, w,
                               (uint16_t *)dx, C, H, W, TRd);
        else
            hipLaunchKernelGGL(dwconv_s2_dgrad_vec_kernel<4>, gridv, dim3(kDwThreads), smv, st, (const uint16_t *)dy, w,
                               (uint16_t *)dx, C, H, W, TRd);
        if (int e = check_launch()) return e;
    } else if (dx) {
        const int TR = pick_rows(H, OW, K, stride, false);
        const size_t sm = sizeof(float) * ((size_t)(TR / stride + K + 1) * OW + K * K);
        dim3 grid(B * C, (H + TR - 1) / TR);
#define DFINE_DWD(TT, KK, SS)                                                                     \
    hipLaunchKernelGGL((dwconv_dgrad_kernel<TT, KK, SS>), grid, dim3(kDwThreads), sm, st, (const TT *)dy, w, (TT *)dx, C, \
                       H, W, OH, OW, K, stride, pad, TR)
#define DFINE_DWD_KS(TT)                                                                          \
    { if (K == 3 && stride == 2) DFINE_DWD(TT, 3, 2); else if (K == 5 && stride == 1) DFINE_DWD(TT, 5, 1);      \
      else if (K == 3 && stride == 1) DFINE_DWD(TT, 3, 1); else DFINE_DWD(TT, 0, 0); }
        if (dtype == DFINE_F32) DFINE_DWD_KS(float) else DFINE_DWD_KS(uint16_t)
#undef DFINE_DWD_KS
#undef DFINE_DWD
        if (int e = check_launch()) return e;
    }
    if (dw_f32) {
        if (!x) return DFINE_E_BADARG;
        // dw_f32 [C, K, K] must be zero-filled by the caller (atomic accumulation over image chunks)
        int per = 1;
        while ((int64_t)C * ((B + per - 1) / per) > 4096 && per < B) per *= 2;
        dim3 grid(C, (B + per - 1) / per);
        if (vec && W / vw <= 64) {
            const int nv = W / vw, ppw = 64 / nv;
            const int cwaves = (C + ppw - 1) / ppw, cblocks = (cwaves + 3) / 4;
            int perb = B;
            while (perb > 1 && (int64_t)cblocks * 4 * ((B + perb - 1) / perb) < 2048) perb = (perb + 1) / 2;
            const int bchunks = (B + perb - 1) / perb;
            int rpc = H;
            while (rpc > 2 * kDwRowsAhead && (int64_t)cblocks * 4 * bchunks * ((H + rpc - 1) / rpc) < 2048) rpc = (rpc + 1) / 2;
            rpc = (rpc + kDwRowsAhead - 1) / kDwRowsAhead * kDwRowsAhead;
            const dim3 gw(cblocks, bchunks, (H + rpc - 1) / rpc);
#define DFINE_WGS(KK, VV) hipLaunchKernelGGL((dwconv_wgrad_s1_stream_kernel<KK, VV>), gw, dim3(kDwThreads), 0, st, \
                                             (const uint16_t *)x, (const uint16_t *)dy, dw_f32, B, C, H, W, perb, rpc)
            if (K == 5) { if (vw == 8) DFINE_WGS(5, 8); else DFINE_WGS(5, 4); }
            else { if (vw == 8) DFINE_WGS(3, 8); else DFINE_WGS(3, 4); }
#undef DFINE_WGS
            return check_launch();
        }
        if (vec) {
            // ~1024 blocks: several images per block amortise the K*K-value block reduction + atomics
            int perv = (int)(((int64_t)B * C + 1023) / 1024);
            if (perv < 1) perv = 1;
            if (perv > B) perv = B;
            per = perv;
            grid = dim3(C, (B + per - 1) / per);
            size_t smv = sizeof(float) * (size_t)(H + K - 1) * (W + 8);
            if (smv < sizeof(float) * (kDwThreads / 64) * K * K) smv = sizeof(float) * (kDwThreads / 64) * K * K;
#define DFINE_WGV(KK, VV) hipLaunchKernelGGL((dwconv_wgrad_s1_vec_kernel<KK, VV>), grid, dim3(kDwThreads), smv, st, \
                                             (const uint16_t *)x, (const uint16_t *)dy, dw_f32, B, C, H, W, per)
            if (K == 5) { if (vw == 8) DFINE_WGV(5, 8); else DFINE_WGV(5, 4); }
            else { if (vw == 8) DFINE_WGV(3, 8); else DFINE_WGV(3, 4); }
#undef DFINE_WGV
            return check_launch();
        }
        if (dw_s2_ok(dtype, H, W, K, stride, pad)) {
            const int nv = W / 8, ppw = 64 / nv;
            const int cwaves = (C + ppw - 1) / ppw, cblocks = (cwaves + 3) / 4;
            // ~2048 waves: images first (no halo rows), then row chunks
            int per = B;
            while (per > 1 && (int64_t)cblocks * 4 * ((B + per - 1) / per) < 2048) per = (per + 1) / 2;
            const int bchunks = (B + per - 1) / per;
            int rpc = OH;
            while (rpc > 2 * kDwRowsAhead && (int64_t)cblocks * 4 * bchunks * ((OH + rpc - 1) / rpc) < 2048) rpc = (rpc + 1) / 2;
            rpc = (rpc + kDwRowsAhead - 1) / kDwRowsAhead * kDwRowsAhead;
            hipLaunchKernelGGL(dwconv_s2_wgrad_stream_kernel, dim3(cblocks, bchunks, (OH + rpc - 1) / rpc), dim3(kDwThreads), 0, st,
                               (const uint16_t *)x, (const uint16_t *)dy, dw_f32, B, C, H, W, per, rpc);
            return check_launch();
        }
        if (dw_s2_ok(dtype, H, W, K, stride, pad)) {
            int perv = (int)(((int64_t)B * C + 1023) / 1024);
            if (perv < 1) perv = 1;
            if (perv > B) perv = B;
            int TRo = kDwThreads / (OW / 4);
            if (TRo < 1) TRo = 1;
            if (TRo > OH) TRo = OH;
            size_t smv = sizeof(float) * (size_t)(2 * TRo + 1) * (W + 8);
            hipLaunchKernelGGL(dwconv_s2_wgrad_vec_kernel, dim3(C, (B + perv - 1) / perv), dim3(kDwThreads), smv, st,
                               (const uint16_t *)x, (const uint16_t *)dy, dw_f32, B, C, H, W, perv, TRo);
            return check_launch();
        }
        const size_t lds_need = sizeof(float) * ((size_t)(H + 2 * pad) * (W + 2 * pad) + (size_t)OH * OW);
        const bool spec = (K == 5 && stride == 1) || (K == 3 && stride == 2) || (K == 3 && stride == 1);
        if (spec && lds_need <= 60 * 1024) {
#define DFINE_WGL(TT, KK, SS)                                                                    \
    hipLaunchKernelGGL((dwconv_wgrad_lds_kernel<TT, KK, SS>), grid, dim3(kDwThreads), lds_need, st, (const TT *)x, \
                       (const TT *)dy, dw_f32, B, C, H, W, OH, OW, pad, per)
#define DFINE_WGL_KS(TT)                                                                         \
    { if (K == 5) DFINE_WGL(TT, 5, 1); else if (stride == 2) DFINE_WGL(TT, 3, 2); else DFINE_WGL(TT, 3, 1); }
            if (dtype == DFINE_F32) DFINE_WGL_KS(float) else DFINE_WGL_KS(uint16_t)
#undef DFINE_WGL_KS
#undef DFINE_WGL
            return check_launch();
        }
#define DFINE_WG(KK, TT)                                                                         \
    hipLaunchKernelGGL((dwconv_wgrad_kernel<TT, KK>), grid, dim3(kDwThreads), 0, st, (const TT *)x, \
                       (const TT *)dy, dw_f32, B, C, H, W, OH, OW, stride, pad, per)
        if (dtype == DFINE_F32) {
            if (K == 3) DFINE_WG(3, float); else if (K == 5) DFINE_WG(5, float);
            else if (K == 1) DFINE_WG(1, float); else if (K == 7) DFINE_WG(7, float);
            else return DFINE_E_BADARG;
        } else {
            if (K == 3) DFINE_WG(3, uint16_t); else if (K == 5) DFINE_WG(5, uint16_t);
            else if (K == 1) DFINE_WG(1, uint16_t); else if (K == 7) DFINE_WG(7, uint16_t);
            else return DFINE_E_BADARG;
        }
#undef DFINE_WG
        if (int e = check_launch()) return e;
    }
    return DFINE_OK;
}

}  // extern "C"
