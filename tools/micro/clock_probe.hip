// clock_probe: does a lone wavefront's serial fp64 chain (the Nose-Hoover sub-step loop's shape) run slower when the launches
// around it stream gigabytes through HBM?  Alternates, in one stream with no host synchronisation, k streaming copies of `mb`
// megabytes with one single-wavefront launch of a dependent chain, and reports for the chain ns per dependent operation,
// cycles per operation (clock64) and the shader clock (clock64 ticks per wall_clock64 tick of 10 ns).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o build_variants/clock_probe && build_variants/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void stream_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = a[i]; v.x += 1.0f; b[i] = v;
    }
}

// the same copy with `flops` dependent fp64 FMAs per element on the way (the integrator's passes do ~100 fp64 operations per slot)
__global__ __launch_bounds__(256) void stream_fma(const double4* __restrict__ a, double4* __restrict__ b, size_t n, int flops) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double4 v = a[i];
        for (int k = 0; k < flops; k += 4) { v.x = fma(v.x, 1.0000001, v.y); v.y = fma(v.y, 0.9999999, v.z); v.z = fma(v.z, 1.0000001, v.w); v.w = fma(v.w, 0.9999999, v.x); }
        b[i] = v;
    }
}

__global__ void chain(double* out, unsigned long long* t, int iters, int slot) {
    double y = 1e-3 * (threadIdx.x + 1), ke = 3.0, acc = 0.0;
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
    if (threadIdx.x == 0) { t[2 * slot] = c1 - c0; t[2 * slot + 1] = w1 - w0; }
    if (acc + ke + y == 12345.678) out[0] = acc;
}

int main() {
    double* out; unsigned long long* t;
    const int reps = 60, iters = 1500;
    CK(hipMalloc((void**)&out, 64)); CK(hipMalloc((void**)&t, 16 * reps));
    float4 *a, *b;
    const size_t max_mb = 800;
    CK(hipMalloc((void**)&a, max_mb << 20)); CK(hipMalloc((void**)&b, max_mb << 20));
    CK(hipMemset(a, 0, max_mb << 20)); CK(hipMemset(b, 0, max_mb << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {(size_t)0, (size_t)16, (size_t)100, (size_t)400, (size_t)800})
    for (int k : {1, 3}) {
        if (mb == 0 && k > 1) continue;
        const size_t n = (mb << 20) / 16;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) {
            for (int j = 0; j < k && mb; j++) hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, 0, a, b, n);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, t, iters, r);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2 * reps];
        CK(hipMemcpy(h, t, 16 * reps, hipMemcpyDeviceToHost));
        double wsum = 0, csum = 0, wmax = 0;
        for (int r = 10; r < reps; r++) { wsum += h[2 * r + 1] * 10.0; csum += h[2 * r]; if (h[2 * r + 1] * 10.0 > wmax) wmax = h[2 * r + 1] * 10.0; }
        const double ops = 7.0 * iters * (reps - 10);
        printf("%d x copy of %4zu MB between chains: %.2f ns per dependent op (worst launch %.2f), %.1f cycles per op, shader clock %.0f MHz; chain %.1f us of %.1f us per round\n",
               mb ? k : 0, mb, wsum / ops, wmax / (7.0 * iters), csum / ops, csum / wsum * 1000.0, wsum / (reps - 10) / 1000.0, ms * 1000.0 / reps);
    }
    for (int flops : {0, 32, 64, 128, 256}) {
        const size_t mb = 800, n = (mb << 20) / 32;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(stream_fma, dim3(2048), dim3(256), 0, 0, (const double4*)a, (double4*)b, n, flops);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, t, iters, r);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2 * reps];
        CK(hipMemcpy(h, t, 16 * reps, hipMemcpyDeviceToHost));
        double wsum = 0, csum = 0;
        for (int r = 10; r < reps; r++) { wsum += h[2 * r + 1] * 10.0; csum += h[2 * r]; }
        const double ops = 7.0 * iters * (reps - 10);
        printf("copy of 800 MB with %3d fp64 FMAs per 32 B between chains: %.2f ns per dependent op, %.1f cycles per op, shader clock %.0f MHz; chain %.1f us of %.1f us per round (copy at %.2f TB/s)\n",
               flops, wsum / ops, csum / ops, csum / wsum * 1000.0, wsum / (reps - 10) / 1000.0, ms * 1000.0 / reps,
               2.0 * (mb << 20) / ((ms * 1000.0 / reps - wsum / (reps - 10) / 1000.0) * 1e-6) / 1e12);
    }
    return 0;
}
