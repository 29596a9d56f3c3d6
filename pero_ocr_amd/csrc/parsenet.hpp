// parsenet.hpp — the two memory-bound ends of the layout network (topology:
// pero_ocr_amd/parsenet_spec.py) around TorchParseNet.get_maps, pero_ocr/layout_engines/torch_parsenet.py:37-58:
//   area_downsample_u8_kernel   cv2.resize(img, (0,0), fx=1/ds, fy=1/ds, INTER_AREA) for an integer factor (:42)
//   parsenet_head_kernel        the 1x1 head conv + ReLU / sigmoid + the crop back to the un-padded size (:52-56)
// The conv layers in between run on conv_igemm_kernel (conv_igemm.hpp: pooling fused into the encoder convs, nearest
// up-sampling + skip concatenation fused into the decoder convs' staging); the first one on conv1_u8_kernel, which
// fuses the zero canvas (:46-47) and the `* (1/255.)` normalisation (:50; a MULTIPLY, unlike the recogniser's / 255.0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pocr {

// OpenCV's INTER_AREA for an integer scale (imgproc resize.cpp, ResizeAreaFast_Invoker, 8-bit): every output pixel is
// the sum of its ds x ds block times the float 1/ds^2, rounded to nearest even; the 2x2 case rounds as (sum + 2) >> 2;
// blocks cut by the image border (source size not a multiple of ds) average the pixels that exist.
// PARITY UNPINNED: cv2 is not installed in the build image, no reference output could be generated.
__global__ __launch_bounds__(256) void area_downsample_u8_kernel(const uint8_t *src, int H, int W, int ds, uint8_t *dst, int Ho, int Wo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Ho * Wo * 3) return;
    const int c = idx % 3, x = (idx / 3) % Wo, y = idx / (3 * Wo);
    const int y0 = y * ds, x0 = x * ds, y1 = min(y0 + ds, H), x1 = min(x0 + ds, W);
    int sum = 0;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) sum += src[((size_t)yy * W + xx) * 3 + c];
    const int cnt = (y1 - y0) * (x1 - x0);
    int v;
    if (cnt == ds * ds) v = ds == 2 ? (sum + 2) >> 2 : (int)rintf((float)sum * (1.0f / (float)(ds * ds)));
    else v = cnt > 0 ? (int)rintf((float)sum / (float)cnt) : 0;
    dst[idx] = (uint8_t)min(max(v, 0), 255);
}

// INTER_AREA for a FRACTIONAL factor (what get_maps_with_optimal_resolution asks for on every page after the first): the host
// builds the separable tap tables (pero_ocr_amd/layout_engines/torch_parsenet._area_taps: output o covers the source interval
// [o * scale, (o + 1) * scale), <= ceil(scale) + 1 taps weighted by the covered length, rows normalised) and this kernel applies
// them in float64 exactly as the host's two sparse products do - rows first, then columns, every product rounded before it is
// added (no fma contraction), taps in ascending order from 0.0 - so device and host agree bit for bit; then rint, clip.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void area_resample_u8_kernel(const uint8_t *src, int H, int W, const double *wy, const int32_t *y0, int ty,
                                                               const double *wx, const int32_t *x0, int tx, uint8_t *dst, int Ho, int Wo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Ho * Wo * 3) return;
    const int c = idx % 3, x = (idx / 3) % Wo, y = idx / (3 * Wo);
    const int ys = y0[y], xs = x0[x];
    double acc = 0.0;
    for (int b = 0; b < tx; ++b) {
        const int xx = min(xs + b, W - 1);
        double col = 0.0;
        for (int a = 0; a < ty; ++a) {
            const int yy = min(ys + a, H - 1);
            const double prod = wy[(size_t)y * ty + a] * (double)src[((size_t)yy * W + xx) * 3 + c];
            col = col + prod;
        }
        const double prod2 = wx[(size_t)x * tx + b] * col;
        acc = acc + prod2;
    }
    const double r = rint(acc);
    dst[idx] = (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}
#pragma clang fp contract(fast)

// y0 [Hp][Wp][64] (decoder output) -> out [h][w][5]: z = W y + b (fmaf chain over the 64 channels), channels 0, 1 ReLU,
// channels 2..4 sigmoid; only the un-padded h x w pixels are written.
__global__ __launch_bounds__(256) void parsenet_head_kernel(const float *y0, int Wp, const float *w5x64, const float *b5, float *out, int h, int w) {
    __shared__ float sw[5 * 64 + 8];
    for (int i = threadIdx.x; i < 5 * 64; i += 256) sw[i] = w5x64[i];
    if (threadIdx.x < 5) sw[320 + threadIdx.x] = b5[threadIdx.x];
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * w) return;
    const int x = idx % w, y = idx / w;
    const float4 *px = reinterpret_cast<const float4 *>(y0 + ((size_t)y * Wp + x) * 64);
    float z[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 v = px[k];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            z[c] = fmaf(v.x, sw[c * 64 + 4 * k + 0], z[c]);
            z[c] = fmaf(v.y, sw[c * 64 + 4 * k + 1], z[c]);
            z[c] = fmaf(v.z, sw[c * 64 + 4 * k + 2], z[c]);
            z[c] = fmaf(v.w, sw[c * 64 + 4 * k + 3], z[c]);
        }
    }
    float *o = out + (size_t)idx * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const float t = z[c] + sw[320 + c];
        o[c] = c < 2 ? fmaxf(t, 0.f) : 1.0f / (1.0f + expf(-t));
    }
}

}  // namespace pocr
