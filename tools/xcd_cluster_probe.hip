// xcd_cluster_probe.hip — what does one step of a RESIDENT recurrence cost when the 16 workgroups that exchange the hidden
// state sit on ONE XCD (block b runs on XCD b % 8) and hand it over through that XCD's L2?
// Per step every workgroup: stores its 1 KB slice of h (16 lines x 16 units), waits for the stores, bumps the cluster's
// counter with an L2-executed atomic, polls the counter with L1-bypassing loads until all 16 members arrived, then reads the
// whole 16 KB of h with L1-bypassing loads.  Prints us per step for 1..64 clusters (= 16 .. 1024 workgroups).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/xcd_cluster_probe tools/xcd_cluster_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void cluster_steps(float *hbuf, unsigned *counters, int nclusters, int steps, int same_xcd, unsigned *errs) {
    const int b = blockIdx.x;
    int cluster, member;
    if (same_xcd) { const int xcd = b & 7, k = b >> 3; cluster = (k / 16) * 8 + xcd; member = k % 16; }
    else { cluster = b / 16; member = b % 16; }           // members spread over the XCDs
    if (cluster >= nclusters) return;
    const int tid = threadIdx.x;
    float *h = hbuf + (size_t)cluster * 2 * 4096;          // [2][16 lines][256 units]
    unsigned *ctr = counters + cluster * 32;               // (own cache line)
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        float *hw = h + (s & 1) * 4096;
        hw[(tid >> 4) * 256 + member * 16 + (tid & 15)] = (float)(s * 4096 + (tid >> 4) * 256 + member * 16 + (tid & 15));       // 1 KB per workgroup; value = f(step, position)
        __builtin_amdgcn_s_waitcnt(0);                                           // stores acknowledged by L2
        __syncthreads();
        if (tid == 0) {
            if (same_xcd) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // executed in this XCD's L2
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            const unsigned want = 16u * (unsigned)(s + 1);
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 22)) {}
            if (spins >= (1 << 22)) atomicAdd(errs, 1u);
            if (!same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // read the whole state of the 16 lines: 16 KB, L1 bypassed (nt loads are served by L2)
        const f32x4 *hr = reinterpret_cast<const f32x4 *>(hw);
        const f32x4 v0 = __builtin_nontemporal_load(hr + tid), v1 = __builtin_nontemporal_load(hr + tid + 256),
                    v2 = __builtin_nontemporal_load(hr + tid + 512), v3 = __builtin_nontemporal_load(hr + tid + 768);      // nt: L1 bypassed
        // every word must be the value its producer wrote in THIS step (checks staleness of every hand-off, L1-warm: the same
        // addresses were read two steps ago)
        {
            const f32x4 vv[4] = {v0, v1, v2, v3};
            unsigned bad = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) bad += vv[q][j] != (float)(s * 4096 + (tid + 256 * q) * 4 + j);
            if (bad) atomicAdd(errs + 3, bad);
        }
        acc = v0[0] + v1[1] + v2[2] + v3[3];
        if (s == steps - 1 && v0[0] != v0[0]) errs[1] = 1;
    }
    if (tid == 0 && acc == -1.f) errs[2] = 1;
}

__global__ void hog(float *p, size_t n, int iters) {        // background load: streams a large buffer (uneven: only some XCDs' worth of blocks)
    for (int it = 0; it < iters; ++it)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.f;
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    float *h; unsigned *ctr, *errs;
    CK(hipMalloc(&h, (size_t)64 * 2 * 4096 * 4)); CK(hipMalloc(&ctr, 64 * 32 * 4)); CK(hipMalloc(&errs, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *big; const size_t nbig = (size_t)256 << 20; CK(hipMalloc(&big, nbig * 4)); CK(hipMemset(big, 0, nbig * 4));
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int load = 0; load < 2; ++load)
    for (int same = 1; same >= 0; --same)
        for (int nc : {1, 2, 6, 8, 16, 32, 64}) {
            CK(hipMemset(ctr, 0, 64 * 32 * 4)); CK(hipMemset(errs, 0, 64)); CK(hipMemset(h, 0, (size_t)64 * 2 * 4096 * 4));
            const int blocks = same ? ((nc + 7) / 8) * 16 * 8 : nc * 16;
            if (load) hipLaunchKernelGGL(hog, dim3(333), dim3(256), 0, s2, big, nbig, 3);     // uneven background load on another stream
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(cluster_steps, dim3(blocks), dim3(256), 0, 0, h, ctr, nc, steps, same, errs);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipDeviceSynchronize());
            unsigned he[4]; CK(hipMemcpy(he, errs, 16, hipMemcpyDeviceToHost));
            printf("%s%s clusters %2d (%4d workgroups): %.2f us per step, stale words %u%s\n", load ? "[under load] " : "", same ? "same-XCD L2 hand-off" : "agent-scope hand-off ", nc, nc * 16,
                   1e3 * ms / steps, he[3], he[0] ? "  (TIMEOUTS!)" : "");
        }
    return 0;
}
