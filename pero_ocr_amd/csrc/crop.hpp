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

// The same remap with the sampling grid generated on the fly from the line's 1-D curves (what the tail of
// get_crop_inputs does on the host with [line_h x width] float64 arrays, crop_engine.py:90-98):
//   g = normal[c] * offset[v] + base[c]        (numpy: one multiply, one add, both rounded)
//   (x, y) = g . R                              (np.dot -> dgemm: second product fused, fma(gy, R1j, gx * R0j))
// and the float32 cast.  curves: per line [4][width] doubles (base_x, base_y, normal_x, normal_y); rows: [line_h]
// offsets; rot: R row-major.  grid_out (optional) receives the float32 grid for tests.
struct CurveLine {
    int64_t curve_off, row_off, rot_off, out_off, grid_off;
    int32_t width;
    int32_t pad_;
};

__device__ __forceinline__ void remap_pixel(const uint8_t *page, int H, int W, int C, float fxp, float fyp, uint8_t *o) {
    const int sx = __float2int_rn(fxp * 32.0f), sy = __float2int_rn(fyp * 32.0f);
    const int ix = max(-32768, min(32767, sx >> 5)), iy = max(-32768, min(32767, sy >> 5));
    const int fx = sx & 31, fy = sy & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    const uint8_t *p00 = page + ((size_t)(y0 ? iy : 0) * W + (x0 ? ix : 0)) * C;
    const uint8_t *p01 = page + ((size_t)(y0 ? iy : 0) * W + (x1 ? ix + 1 : 0)) * C;
    const uint8_t *p10 = page + ((size_t)(y1 ? iy + 1 : 0) * W + (x0 ? ix : 0)) * C;
    const uint8_t *p11 = page + ((size_t)(y1 ? iy + 1 : 0) * W + (x1 ? ix + 1 : 0)) * C;
    for (int c = 0; c < C; ++c) {
        const int v = w00 * (y0 && x0 ? p00[c] : 0) + w01 * (y0 && x1 ? p01[c] : 0) + w10 * (y1 && x0 ? p10[c] : 0) +
                      w11 * (y1 && x1 ? p11[c] : 0);
        o[c] = (uint8_t)min(255, max(0, (v + (1 << 14)) >> 15));
    }
}

__global__ __launch_bounds__(256) void remap_curves_u8_kernel(const uint8_t *page, int H, int W, int C, const double *curves,
                                                              const double *rows, const double *rot, const CurveLine *lines,
                                                              int line_h, uint8_t *out, float *grid_out) {
    const CurveLine ln = lines[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= line_h * ln.width) return;
    const int v = idx / ln.width, c = idx % ln.width;
    const double *cv = curves + ln.curve_off;
    const double off = rows[ln.row_off + v];
    const double gx = __dadd_rn(__dmul_rn(cv[2 * (size_t)ln.width + c], off), cv[c]);
    const double gy = __dadd_rn(__dmul_rn(cv[3 * (size_t)ln.width + c], off), cv[(size_t)ln.width + c]);
    const double *R = rot + ln.rot_off;
    const float x = (float)__fma_rn(gy, R[2], __dmul_rn(gx, R[0]));
    const float y = (float)__fma_rn(gy, R[3], __dmul_rn(gx, R[1]));
    if (grid_out) {
        grid_out[ln.grid_off + 2 * (size_t)idx] = x;
        grid_out[ln.grid_off + 2 * (size_t)idx + 1] = y;
    }
    remap_pixel(page, H, W, C, x, y, out + ln.out_off + (size_t)idx * C);
}

}  // namespace pocr
