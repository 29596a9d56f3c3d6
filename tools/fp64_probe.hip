// Are the float64 primitives the cropper relies on correctly rounded on gfx950?  Compares device results with the host's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#pragma clang fp contract(off)
__device__ __forceinline__ double f64_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double f64_add(double a, double b) { return a + b; }
#pragma clang fp contract(fast)
__global__ void k(const double *a, const double *b, double *o, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    o[i] = __ddiv_rn(a[i], b[i]);
    o[n + i] = __dsqrt_rn(a[i]);
    o[2 * n + i] = __dmul_rn(a[i], b[i]);
    o[3 * n + i] = __dadd_rn(a[i], b[i]);
    o[4 * n + i] = a[i] / b[i];
    o[5 * n + i] = sqrt(a[i]);
    o[6 * n + i] = __dadd_rn(__dmul_rn(a[i], b[i]), a[i]);
    o[7 * n + i] = f64_add(f64_mul(a[i], b[i]), a[i]);
    o[8 * n + i] = f64_add(f64_mul(a[i], b[i]), f64_mul(b[i], b[i]));
}
int main() {
    const int n = 1 << 20;
    std::vector<double> a(n), b(n), o(9 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0.001, 3000.0);
    for (int i = 0; i < n; ++i) { a[i] = u(g); b[i] = u(g); }
    double *da, *db, *dd;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dd, 9 * n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dd, n);
    hipMemcpy(o.data(), dd, 9 * n * 8, hipMemcpyDeviceToHost);
    long bad[9] = {0};
    for (int i = 0; i < n; ++i) {
        volatile double m = a[i] * b[i];
        volatile double m2 = b[i] * b[i];
        double ref[9] = {a[i] / b[i], std::sqrt(a[i]), a[i] * b[i], a[i] + b[i], a[i] / b[i], std::sqrt(a[i]), m + a[i], m + a[i], m + m2};
        for (int q = 0; q < 9; ++q) bad[q] += o[q * n + i] != ref[q];
    }
    printf("mismatches of %d: __ddiv_rn %ld  __dsqrt_rn %ld  __dmul_rn %ld  __dadd_rn %ld  a/b %ld  sqrt %ld  mul-then-add via __d*_rn %ld  via contract(off) helpers %ld  two products %ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7], bad[8]);
    return 0;
}
