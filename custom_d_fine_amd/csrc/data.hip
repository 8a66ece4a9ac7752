// (f3) device-side data path, first pieces: the geometric augmentations of the reference's training samples on the GPU.
// Reference (host, per DataLoader worker: OpenCV + numpy): CustomDataset._load_mosaic (src/dl/dataset.py:258-377) =
// 4 x cv2.resize into a 2H x 2W canvas at get_mosaic_coordinate (src/dl/utils.py:392-414), random_affine (utils.py:325-389) =
// cv2.warpAffine of the canvas + the affine map of the box corners, clipping and the box_candidates filter (utils.py:283-295).
//   mosaic_place_kernel  one source frame -> its canvas region: OpenCV's 8-bit INTER_LINEAR resize arithmetic (the same
//                        restatement as preprocess_kernel in postproc.hip) fused with the crop / placement copy;
//   warp_affine_kernel   canvas -> target frame: cv2.warpAffine(INTER_LINEAR, constant border) restated - inverse map in
//                        10-bit fixed point, 1/32-pixel sub-positions, bilinear weights from the 2^15-scaled table
//                        (imgproc/imgwarp.cpp; opencv-python is a pip dependency of the reference, absent here: PARITY
//                        UNPINNED against cv2 itself, checked by properties);
//   affine_boxes_kernel  xyxy boxes through the same matrix: corner map, min / max, clip, candidate filter - plain fp32
//                        arithmetic in the reference's operation order (pinned by oracle/np_ref.py against hand cases).
// uint8 HWC images, one thread per output pixel: HBM-bound byte work, nothing to tile.
#include "common.h"

namespace dfine {

__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int *s0, int *a0, int *a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    *s0 = s;
    *a0 = (int)lrintf((1.f - f) * 2048.f);
    *a1 = (int)lrintf(f * 2048.f);
}

// canvas [Hc, Wc, 3] region [ly1, ly2) x [lx1, lx2)  <-  resize(src [Hs, Ws, 3] -> [rh, rw]) [sy1 + .., sx1 + ..]
__global__ __launch_bounds__(256) void mosaic_place_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ canvas, int Hs, int Ws, int rh,
                                                           int rw, int Wc, int lx1, int ly1, int lx2, int ly2, int sx1, int sy1) {
    const int w = lx2 - lx1, h = ly2 - ly1;
    const double sx = (double)Ws / rw, sy = (double)Hs / rh;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int yy = i / w, xx = i - yy * w;
        const int rx = sx1 + xx, ry = sy1 + yy;                 // pixel of the resized frame
        uint8_t out[3];
        if (rw == Ws && rh == Hs) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = src[((int64_t)ry * Ws + rx) * 3 + c];
        } else {
            int x0, ax0, ax1, y0, ay0, ay1;
            lin_coef(rx, sx, Ws, &x0, &ax0, &ax1);
            lin_coef(ry, sy, Hs, &y0, &ay0, &ay1);
            const int x1 = min(x0 + 1, Ws - 1), y1 = min(y0 + 1, Hs - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = src[((int64_t)y0 * Ws + x0) * 3 + c] * ax0 + src[((int64_t)y0 * Ws + x1) * 3 + c] * ax1;
                const int r1 = src[((int64_t)y1 * Ws + x0) * 3 + c] * ax0 + src[((int64_t)y1 * Ws + x1) * 3 + c] * ax1;
                out[c] = (uint8_t)((((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2);
            }
        }
        uint8_t *o = canvas + ((int64_t)(ly1 + yy) * Wc + lx1 + xx) * 3;
        o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
    }
}

// dst [Hd, Wd, 3] = warpAffine(src [Hs, Ws, 3], M (forward, 2 x 3), INTER_LINEAR, BORDER_CONSTANT value).  minv = inverse map.
__global__ __launch_bounds__(256) void warp_affine_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int Hs, int Ws, int Hd,
                                                          int Wd, double m00, double m01, double m02, double m10, double m11, double m12,
                                                          int border) {
    constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 1 << INTER_BITS, ROUND_DELTA = AB_SCALE / TAB / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hd * Wd; i += gridDim.x * 256) {
        const int y = i / Wd, x = i - y * Wd;
        // cv::warpAffine: adelta / bdelta per column, X0 / Y0 per row, all in AB_SCALE fixed point
        const int adelta = (int)lrint(m00 * x * AB_SCALE), bdelta = (int)lrint(m10 * x * AB_SCALE);
        const int X0 = (int)lrint((m01 * y + m02) * AB_SCALE) + ROUND_DELTA, Y0 = (int)lrint((m11 * y + m12) * AB_SCALE) + ROUND_DELTA;
        const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
        const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS, fx = X & (TAB - 1), fy = Y & (TAB - 1);
        // bilinear table weights: (1 - f/32, f/32) per axis, products scaled to 2^15 with the table's rounding
        const float wx1 = (float)fx / TAB, wy1 = (float)fy / TAB;
        int w[4] = {(int)lrintf((1.f - wy1) * (1.f - wx1) * 32768.f), (int)lrintf((1.f - wy1) * wx1 * 32768.f),
                    (int)lrintf(wy1 * (1.f - wx1) * 32768.f), (int)lrintf(wy1 * wx1 * 32768.f)};
        // the table is normalised to sum 2^15 by adjusting its largest entry
        const int diff = 32768 - (w[0] + w[1] + w[2] + w[3]);
        int kmax = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (w[k] > w[kmax]) kmax = k;
        w[kmax] += diff;
        uint8_t *o = dst + (int64_t)i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = sx + (k & 1), py = sy + (k >> 1);
                v[k] = (px >= 0 && px < Ws && py >= 0 && py < Hs) ? (int)src[((int64_t)py * Ws + px) * 3 + c] : border;
            }
            const int acc = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
            o[c] = (uint8_t)min(max((acc + (1 << 14)) >> 15, 0), 255);
        }
    }
}

// boxes [N, 4] xyxy (canvas pixels) -> out [N, 4] in the target frame, keep [N] (utils.py:343-377, 283-295):
// corners (x1,y1) (x2,y2) (x1,y2) (x2,y1) through M, min / max, clip to [0, tw] x [0, th], then
// keep = w2 > 2 && h2 > 2 && w2 h2 / (w1 h1 s^2 + eps) > area_thr && max(w2 / (h2 + eps), h2 / (w2 + eps)) < 20
__global__ void affine_boxes_kernel(const float *__restrict__ boxes, float *__restrict__ out, uint8_t *__restrict__ keep, int N, float m00,
                                    float m01, float m02, float m10, float m11, float m12, float scale, float tw, float th,
                                    float area_thr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float x1 = boxes[i * 4], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
    const float cx[4] = {x1, x2, x1, x2}, cy[4] = {y1, y2, y2, y1};
    float nx0 = 3.4e38f, ny0 = 3.4e38f, nx1 = -3.4e38f, ny1 = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float X = cx[k] * m00 + cy[k] * m01 + m02, Y = cx[k] * m10 + cy[k] * m11 + m12;
        nx0 = fminf(nx0, X); nx1 = fmaxf(nx1, X); ny0 = fminf(ny0, Y); ny1 = fmaxf(ny1, Y);
    }
    nx0 = fminf(fmaxf(nx0, 0.f), tw); nx1 = fminf(fmaxf(nx1, 0.f), tw);
    ny0 = fminf(fmaxf(ny0, 0.f), th); ny1 = fminf(fmaxf(ny1, 0.f), th);
    out[i * 4] = nx0; out[i * 4 + 1] = ny0; out[i * 4 + 2] = nx1; out[i * 4 + 3] = ny1;
    const float eps = 1e-16f;
    const float w1 = (x2 - x1) * scale, h1 = (y2 - y1) * scale, w2 = nx1 - nx0, h2 = ny1 - ny0;
    const float ar = fmaxf(w2 / (h2 + eps), h2 / (w2 + eps));
    keep[i] = (w2 > 2.f && h2 > 2.f && w2 * h2 / (w1 * h1 + eps) > area_thr && ar < 20.f) ? 1 : 0;
}

}  // namespace dfine

using namespace dfine;

extern "C" {

int dfine_mosaic_place_u8(const uint8_t *src, uint8_t *canvas, int Hs, int Ws, int rh, int rw, int Hc, int Wc, int lx1, int ly1, int lx2,
                          int ly2, int sx1, int sy1, void *stream) {
    if (lx2 <= lx1 || ly2 <= ly1) return DFINE_OK;
    if (!src || !canvas || Hs < 1 || Ws < 1 || rh < 1 || rw < 1 || lx1 < 0 || ly1 < 0 || lx2 > Wc || ly2 > Hc || sx1 < 0 || sy1 < 0 ||
        sx1 + (lx2 - lx1) > rw || sy1 + (ly2 - ly1) > rh)
        return DFINE_E_BADARG;
    const int n = (lx2 - lx1) * (ly2 - ly1);
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mosaic_place_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, canvas, Hs, Ws, rh, rw, Wc, lx1, ly1, lx2,
                       ly2, sx1, sy1);
    return check_launch();
}

// m: the FORWARD 2 x 3 matrix (as passed to cv2.warpAffine without WARP_INVERSE_MAP); inverted here in double like OpenCV.
int dfine_warp_affine_u8(const uint8_t *src, uint8_t *dst, int Hs, int Ws, int Hd, int Wd, const double *m, int border, void *stream) {
    if (!src || !dst || !m || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1) return DFINE_E_BADARG;
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0 ? 1.0 / D : 0.0;
    const double A11 = m[4] * D, A22 = m[0] * D;
    const double i00 = A11, i01 = -m[1] * D, i10 = -m[3] * D, i11 = A22;
    const double b1 = -i00 * m[2] - i01 * m[5], b2 = -i10 * m[2] - i11 * m[5];
    int blocks = (Hd * Wd + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(warp_affine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, Hs, Ws, Hd, Wd, i00, i01, b1, i10, i11, b2,
                       border);
    return check_launch();
}

int dfine_affine_boxes(const float *boxes, float *out, uint8_t *keep, int N, const float *m, float scale, float target_w, float target_h,
                       float area_thr, void *stream) {
    if (N == 0) return DFINE_OK;
    if (!boxes || !out || !keep || !m || N < 0) return DFINE_E_BADARG;
    hipLaunchKernelGGL(affine_boxes_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, boxes, out, keep, N, m[0], m[1], m[2], m[3],
                       m[4], m[5], scale, target_w, target_h, area_thr);
    return check_launch();
}

}  // extern "C"
