// chain_probe: how fast does ONE wavefront run a serial fp64 dependency chain (the shape of the Nose-Hoover sub-step loop:
// 7 dependent FMAs per sub-step) when 1, 256, 768 ... work-groups do the same at once?  Reports ns per dependent operation
// and the shader clock (clock64 ticks per wall_clock64 tick).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/chain_probe.hip -o build_variants/chain_probe && build_variants/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void chain(double* out, unsigned long long* t, int iters, int waves_chaining, int others_spin) {
    const int wave = threadIdx.x >> 6;
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    double y = 1e-3 * (threadIdx.x + 1), ke = 3.0, acc = 0.0;
    if (wave < waves_chaining) {
        const unsigned long long c0 = clock64(), w0 = wall_clock64();
        for (int i = 0; i < iters; i++) {                 // 7 dependent operations per round
            double p = fma(y, 0.5, 1.0);
            p = fma(p, y, 1.0);
            p = fma(p, y, 1.0);
            ke = ke * p;
            const double edd = fma(ke, 1e-3, -2.9e-3);
            y = fma(edd, -1e-4, y);
            acc = fma(edd, 0.25, acc);
        }
        const unsigned long long c1 = clock64(), w1 = wall_clock64();
        if ((threadIdx.x & 63) == 0 && wave == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
        if (threadIdx.x == 0) done = 1;
    } else if (others_spin) {
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (acc + ke + y == 12345.678) out[0] = acc;
}

int main() {
    double* out; unsigned long long* t;
    CK(hipMalloc((void**)&out, 64)); CK(hipMalloc((void**)&t, 16 * 4096));
    const int iters = 4000;
    for (int spin = 0; spin < 2; spin++)
    for (int wc : {1, 2, 4})
    for (int grid : {1, 256, 512, 768, 1024}) {
        unsigned long long h[2 * 4096];
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, out, t, iters, wc, spin);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, t, 16 * grid, hipMemcpyDeviceToHost));
        double cmin = 1e30, cmax = 0, wsum = 0, csum = 0;
        for (int b = 0; b < grid; b++) { const double ns = h[2 * b + 1] * 10.0; wsum += ns; csum += h[2 * b]; cmin = ns < cmin ? ns : cmin; cmax = ns > cmax ? ns : cmax; }
        printf("others %s  chaining waves/WG %d  work-groups %4d : %.2f ns per dependent op (min %.2f max %.2f), %.1f cycles per op, shader clock %.0f MHz\n",
               spin ? "spin " : "sleep", wc, grid, wsum / grid / (7.0 * iters), cmin / (7.0 * iters), cmax / (7.0 * iters),
               csum / grid / (7.0 * iters), csum / wsum * 1000.0);
    }
    return 0;
}
