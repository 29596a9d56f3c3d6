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

// float64 operations that round on their own.  hipcc compiles device code with -ffp-contract=fast and HIP's __dmul_rn /
// __dadd_rn are plain `*` / `+` in a header compiled that way: a product feeding a sum becomes v_fma_f64 (measured:
// tools/fp64_probe.hip, 25 % of random mul-then-add results differ from the host's).  numpy and scipy's C++ round
// every operation, so the cropper's float64 arithmetic goes through these (no contract flag -> the backend cannot fuse).
#pragma clang fp contract(off)
__device__ __forceinline__ double f64_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double f64_add(double a, double b) { return a + b; }
__device__ __forceinline__ double f64_sub(double a, double b) { return a - b; }
__device__ __forceinline__ double f64_div(double a, double b) { return a / b; }
__device__ __forceinline__ double f64_sqrt(double a) { return __builtin_sqrt(a); }
#pragma clang fp contract(fast)

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
    const double gx = f64_add(f64_mul(cv[2 * (size_t)ln.width + c], off), cv[c]);
    const double gy = f64_add(f64_mul(cv[3 * (size_t)ln.width + c], off), cv[(size_t)ln.width + c]);
    const double *R = rot + ln.rot_off;
    const float x = (float)__fma_rn(gy, R[2], f64_mul(gx, R[0]));
    const float y = (float)__fma_rn(gy, R[3], f64_mul(gx, R[1]));
    if (grid_out) {
        grid_out[ln.grid_off + 2 * (size_t)idx] = x;
        grid_out[ln.grid_off + 2 * (size_t)idx + 1] = y;
    }
    remap_pixel(page, H, W, C, x, y, out + ln.out_off + (size_t)idx * C);
}


// ---- the whole of get_crop_inputs on the device ------------------------------------------------------------------
// The host keeps what is a handful of scalar operations per LINE (integer baseline, rotation, the interpolant's
// coefficients: a banded 4-point solve); everything per COLUMN and per PIXEL of pero_ocr/core/crop_engine.py:73-98
// runs here in float64, operation for operation (every product and sum rounded separately: scipy's C++ and numpy do
// not contract to FMA), so the sampling grid is bit-identical to the reference's:
//   arc kernel      x = arange(x_min, x_max); y = f(x); L = cumsum(hypot-by-hand of the unit steps)[-1]     (:73-75)
//                   (sequential sum - cumsum's rounding order is part of the contract), width = int(L * zoom)
//   column kernel   t = linspace(0, L, width); bx = the reference's never-advancing reverse_line_mapping (:101-111:
//                   every sample interpolated on the wrap-around pair), by = f(bx), normal from f(bx + 0.1) (:77-89)
//   pixel kernel    rows = linspace(-up, down, line_h); grid = normal * row + base; . R; float32; OpenCV remap (:90-99,146-163)
// f = scipy.interpolate.interp1d(kind="cubic") = a cubic B-spline evaluated with scipy's de Boor recurrence
// (scipy/interpolate/src/__fitpack.h _deBoor_D + _evaluate_spline, pinned version 1.15.3), or np.poly1d (Horner).
struct CropSpec {
    double x_min, x_max;    // np.arange(x_min, x_max)
    double lo, hi;          // interp1d raises outside [lo, hi] (bounds_error): the line then takes the reference's failure path
    double zoom;            // target_height / (above + below)
    double above, below;    // scaled line heights: rows = linspace(-above, below, line_h)
    double rot[4];          // R, row-major
    int32_t mode;           // 0 cubic B-spline (knots + coefficients), 1 polynomial (np.poly1d coefficients, highest power first)
    int32_t n_coef;
    int32_t coef_off, knot_off;
    int32_t n_x;            // len(arange(x_min, x_max))
    int32_t pad_;
};
struct CropState {          // device-side results of the arc kernel, one per line
    double L;               // arc[-1]
    int32_t width;          // int(L * zoom); 0 = empty grid
    int32_t status;         // 0 ok, 1 evaluation outside the interpolant's domain / non-finite
    int64_t curve_off, out_off, grid_off;   // filled by the host between the kernels
};

__device__ __forceinline__ double crop_eval_f(const CropSpec &sp, const double *__restrict__ knots, const double *__restrict__ coefs, double x) {
    const double *c = coefs + sp.coef_off;
    if (sp.mode == 1) {                                  // np.polyval: y = 0; for pv in p: y = y * x + pv
        double y = 0.0;
        for (int i = 0; i < sp.n_coef; ++i) y = f64_add(f64_mul(y, x), c[i]);
        return y;
    }
    const double *t = knots + sp.knot_off;
    const int n = sp.n_coef;                             // len(t) - k - 1
    // find_interval: the l in [k, n-1] with t[l] <= x < t[l+1] (x == t[n] -> n-1); same result as scipy's linear walk
    int lo = 3, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid] <= x) lo = mid; else hi = mid - 1;
    }
    const int ell = lo;
    double h[4] = {1.0, 0.0, 0.0, 0.0}, hh[4];
#pragma unroll
    for (int j = 1; j <= 3; ++j) {
#pragma unroll
        for (int q = 0; q < 3; ++q) if (q < j) hh[q] = h[q];
        h[0] = 0.0;
#pragma unroll
        for (int q = 1; q <= 3; ++q) {
            if (q > j) continue;
            const double xb = t[ell + q], xa = t[ell + q - j];
            if (xb == xa) { h[q] = 0.0; continue; }
            const double w = f64_div(hh[q - 1], f64_sub(xb, xa));
            h[q - 1] = f64_add(h[q - 1], f64_mul(w, f64_sub(xb, x)));
            h[q] = f64_mul(w, f64_sub(x, xa));
        }
    }
    double out = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) out = f64_add(out, f64_mul(c[ell + a - 3], h[a]));
    return out;
}

constexpr int CROP_ARC_TILE = 2048;

// np.arange(start, stop)[i] for float64 (numpy's DOUBLE_fill): the first two elements are start and start + 1, the rest
// start + i * delta with delta = (start + 1) - start - which is not 1 when start + 1 leaves start's binade.
__device__ __forceinline__ double crop_arange(double start, int i) {
    const double delta = f64_sub(f64_add(start, 1.0), start);
    return f64_add(start, f64_mul((double)i, delta));
}

// one workgroup per line
__global__ __launch_bounds__(256) void crop_arc_kernel(const CropSpec *specs, const double *knots, const double *coefs, CropState *state) {
    __shared__ double s_y[CROP_ARC_TILE + 1];
    __shared__ double s_seg[CROP_ARC_TILE];
    __shared__ double s_acc;
    __shared__ int s_bad;
    const CropSpec sp = specs[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid == 0) { s_acc = 0.0; s_bad = 0; }
    __syncthreads();
    const int nseg = sp.n_x - 1;
    for (int base = 0; base < nseg; base += CROP_ARC_TILE) {
        const int cnt = min(CROP_ARC_TILE, nseg - base);
        for (int i = tid; i <= cnt; i += 256) {
            const double x = crop_arange(sp.x_min, base + i);
            if (sp.mode == 0 && !(x >= sp.lo && x <= sp.hi)) s_bad = 1;
            s_y[i] = crop_eval_f(sp, knots, coefs, x);
        }
        __syncthreads();
        for (int i = tid; i < cnt; i += 256) {
            const double x0 = crop_arange(sp.x_min, base + i), x1 = crop_arange(sp.x_min, base + i + 1);
            const double dx = f64_sub(x0, x1), dy = f64_sub(s_y[i], s_y[i + 1]);
            s_seg[i] = f64_sqrt(f64_add(f64_mul(dx, dx), f64_mul(dy, dy)));
        }
        __syncthreads();
        if (tid == 0) {                                  // np.cumsum: strictly sequential
            double acc = s_acc;
            int i = 0;
            if (base == 0) { acc = s_seg[0]; i = 1; }
            for (; i + 8 <= cnt; i += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = s_seg[i + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = f64_add(acc, v[q]);
            }
            for (; i < cnt; ++i) acc = f64_add(acc, s_seg[i]);
            s_acc = acc;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double L = nseg > 0 ? s_acc : 0.0;
        const double wd = f64_mul(L, sp.zoom);
        CropState st = state[blockIdx.x];
        st.L = L;
        const bool finite = wd == wd && fabs(wd) < 2.0e9;
        st.status = (s_bad || !finite || sp.n_x < 1) ? 1 : 0;
        st.width = (finite && wd > 0.0 && !st.status) ? (int)wd : 0;       // int(): truncation
        state[blockIdx.x] = st;
    }
}

// grid (ceil(w_max / 256), n_lines): the per-column curves of one line -> curves[4][width] (base_x, base_y, normal_x, normal_y)
__global__ __launch_bounds__(256) void crop_columns_kernel(const CropSpec *specs, const double *knots, const double *coefs,
                                                           CropState *state, double *curves) {
    const CropSpec sp = specs[blockIdx.y];
    const CropState st = state[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x, n = st.width;
    if (j >= n || st.status) return;
    const double L = st.L;
    double t;
    if (n > 1) {
        const double step = f64_div(L, (double)(n - 1));                // np.linspace: arange(num) * step (+ start = 0), last = stop
        t = (j == n - 1) ? L : f64_mul((double)j, step);
    } else {
        t = f64_mul(0.0, L);
    }
    const double d = f64_sub(0.0, L);                                   // forward_mapping[0] - forward_mapping[-1]
    const double da = f64_div(f64_sub(t, L), d);
    const double x_first = sp.x_min, x_last = crop_arange(sp.x_min, sp.n_x - 1);
    const double bx = f64_add(f64_mul(f64_sub(1.0, da), x_last), f64_mul(da, x_first));
    const double bx2 = f64_add(bx, 0.1);
    if (sp.mode == 0 && !(bx >= sp.lo && bx <= sp.hi && bx2 >= sp.lo && bx2 <= sp.hi)) { state[blockIdx.y].status = 1; return; }
    const double by = crop_eval_f(sp, knots, coefs, bx);
    const double ddy = f64_sub(by, crop_eval_f(sp, knots, coefs, bx2));
    const double len = f64_sqrt(f64_add(f64_mul(0.1, 0.1), f64_mul(ddy, ddy)));
    double *cv = curves + st.curve_off;
    cv[j] = bx;
    cv[(size_t)n + j] = by;
    cv[2 * (size_t)n + j] = f64_div(-ddy, len);
    cv[3 * (size_t)n + j] = f64_div(0.1, len);
}

// grid (ceil(line_h * w_max / 256), n_lines): grid generation + remap from the device-side curves
__global__ __launch_bounds__(256) void remap_spec_u8_kernel(const uint8_t *page, int H, int W, int C, const CropSpec *specs,
                                                            const CropState *state, const double *curves, int line_h,
                                                            uint8_t *out, float *grid_out) {
    const CropState st = state[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (st.status || idx >= line_h * st.width) return;
    const CropSpec &sp = specs[blockIdx.y];
    const int v = idx / st.width, c = idx % st.width;
    // rows = np.linspace(-above, below, line_h)
    const double start = -sp.above, delta = f64_sub(sp.below, start);
    double off;
    if (line_h > 1) {
        const double step = f64_div(delta, (double)(line_h - 1));
        off = (v == line_h - 1) ? sp.below
                                : (step == 0.0 ? f64_add(f64_mul(f64_div((double)v, (double)(line_h - 1)), delta), start)
                                               : f64_add(f64_mul((double)v, step), start));
    } else {
        off = f64_add(f64_mul(0.0, delta), start);
    }
    const double *cv = curves + st.curve_off;
    const size_t w = (size_t)st.width;
    const double gx = f64_add(f64_mul(cv[2 * w + c], off), cv[c]);
    const double gy = f64_add(f64_mul(cv[3 * w + c], off), cv[w + c]);
    const float x = (float)__fma_rn(gy, sp.rot[2], f64_mul(gx, sp.rot[0]));
    const float y = (float)__fma_rn(gy, sp.rot[3], f64_mul(gx, sp.rot[1]));
    if (grid_out) {
        grid_out[st.grid_off + 2 * (size_t)idx] = x;
        grid_out[st.grid_off + 2 * (size_t)idx + 1] = y;
    }
    remap_pixel(page, H, W, C, x, y, out + st.out_off + (size_t)idx * C);
}

}  // namespace pocr
