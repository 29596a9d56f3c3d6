// crop.hpp — line cropper remap.  Replaces the cv2.remap(img, map_x, map_y,
// INTER_LINEAR, BORDER_CONSTANT) call of EngineLineCropper.fast_remap, pero_ocr/core/crop_engine.py:146-163
// (the reference's crop-to-bounding-box variant, :156-162, gives the same pixels: integer shifts of the float32
// coordinates are exact and taps outside the sub-image carry weight 0).
// Arithmetic = OpenCV's 8-bit bilinear remap (modules/imgproc/src/imgwarp.cpp): coordinates rounded to 1/32 pixel
// (round half to even), the four weights (32-fx)(32-fy)32 ... fx fy 32 sum to 2^15 exactly, result
// (sum + 2^14) >> 15, taps outside the image read as 0.  OpenCV is not installed in the build image, so this
// kernel is pinned against the restatement in oracle/crop_oracle.py only (parity with cv2 itself: unpinned).
// One thread per output pixel (all channels): a pure gather, HBM/L2-bound; 8 B of coordinates in, C bytes out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

struct CropLine {
    int64_t coord_off;   // first float of this line's [line_h][width][2] (x, y) grid
    int64_t out_off;     // first byte of this line's [line_h][width][C] crop
    int32_t width;
    int32_t pad_;
};

__global__ __launch_bounds__(256) void remap_u8_kernel(const uint8_t *page, int H, int W, int C, const float *coords,
                                                       const CropLine *lines, int line_h, uint8_t *out) {
    const CropLine ln = lines[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= line_h * ln.width) return;
    const float2 xy = reinterpret_cast<const float2 *>(coords + ln.coord_off)[idx];
    const int sx = __float2int_rn(xy.x * 32.0f), sy = __float2int_rn(xy.y * 32.0f);
    const int ix = max(-32768, min(32767, sx >> 5)), iy = max(-32768, min(32767, sy >> 5));      // saturate_cast<short>
    const int fx = sx & 31, fy = sy & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    const uint8_t *p00 = page + ((size_t)(y0 ? iy : 0) * W + (x0 ? ix : 0)) * C;
    const uint8_t *p01 = page + ((size_t)(y0 ? iy : 0) * W + (x1 ? ix + 1 : 0)) * C;
    const uint8_t *p10 = page + ((size_t)(y1 ? iy + 1 : 0) * W + (x0 ? ix : 0)) * C;
    const uint8_t *p11 = page + ((size_t)(y1 ? iy + 1 : 0) * W + (x1 ? ix + 1 : 0)) * C;
    uint8_t *o = out + ln.out_off + (size_t)idx * C;
    for (int c = 0; c < C; ++c) {
        const int v = w00 * (y0 && x0 ? p00[c] : 0) + w01 * (y0 && x1 ? p01[c] : 0) + w10 * (y1 && x0 ? p10[c] : 0) +
                      w11 * (y1 && x1 ? p11[c] : 0);
        o[c] = (uint8_t)min(255, max(0, (v + (1 << 14)) >> 15));
    }
}

}  // namespace pocr
