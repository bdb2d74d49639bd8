// read_probe: what does a READ-ONLY streaming pass reach on this box?  The denominator for the kick+KE pass (V r, F r:
// 56 B/slot mixed = 280 MB at 5 M slots), which stores nothing -- a device copy (r + w) is the wrong yardstick for it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/read_probe.hip -o build_variants/read_probe && build_variants/read_probe
// Two access shapes, both summed into one double per work-group (so the loads cannot be dropped):
//   flat   one 280 MB array, 32 bytes per lane per load (double4), persistent grid
//   kick   the pass's own shape: velm double4 [N] + meta u32 [N] + three int64 force planes [padded]
// each over grids of {1, 2, 3, 4, 5, 6, 8} work-groups per compute unit and {1, 2, 4} loads in flight per lane.  Every timed
// launch reads a different one of 4 copies of the data (1.1 GB apart in total), so nothing comes from the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double wsum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void read_flat(const double4* __restrict__ a, size_t n, double* out) {
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        double4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = a[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) s += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n; i += stride) { const double4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kick(const double4* __restrict__ velm, const unsigned* __restrict__ meta,
                                                 const long long* __restrict__ force, size_t n, size_t padded, double* out) {
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        double4 v[UNROLL]; unsigned m[UNROLL]; long long fx[UNROLL], fy[UNROLL], fz[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const size_t j = i + k * stride;
            v[k] = velm[j]; m[k] = meta[j]; fx[k] = force[j]; fy[k] = force[j + padded]; fz[k] = force[j + 2 * padded];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const double c = 1e-12 * v[k].w;
            const double x = v[k].x + c * (double)fx[k], y = v[k].y + c * (double)fy[k], z = v[k].z + c * (double)fz[k];
            s += (x * x + y * y + z * z) * (double)(m[k] & 3u);
        }
    }
    for (; i < n; i += stride) { const double4 v = velm[i]; s += v.x * (double)force[i] + (double)meta[i]; }
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}

int main(int argc, char** argv) {
    const size_t N = argc > 1 ? (size_t)atol(argv[1]) : 5000000;
    const int COPIES = 4, REPS = 12;
    int ncu = 256;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount;
    const size_t flat_n = N * 56 / 32;                      // the same 280 MB as double4 elements
    std::vector<double4*> flat(COPIES), velm(COPIES); std::vector<unsigned*> meta(COPIES); std::vector<long long*> force(COPIES);
    for (int c = 0; c < COPIES; c++) {
        CK(hipMalloc(&flat[c], flat_n * 32)); CK(hipMemset(flat[c], 0, flat_n * 32));
        CK(hipMalloc(&velm[c], N * 32)); CK(hipMemset(velm[c], 0, N * 32));
        CK(hipMalloc(&meta[c], N * 4)); CK(hipMemset(meta[c], 0, N * 4));
        CK(hipMalloc(&force[c], N * 24)); CK(hipMemset(force[c], 0, N * 24));
    }
    double* out; CK(hipMalloc(&out, 8 * 4096)); CK(hipMemset(out, 0, 8 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("read-only stream, %zu slots: flat = %.1f MB as double4; kick = velm %.1f + meta %.1f + force %.1f MB (algorithmic %.1f MB without meta)\n",
           N, flat_n * 32 / 1e6, N * 32 / 1e6, N * 4 / 1e6, N * 24 / 1e6, N * 56 / 1e6);
    const int per_cu[] = {1, 2, 3, 4, 5, 6, 8};
    for (int shape = 0; shape < 2; shape++) {
        for (int u = 0; u < 3; u++) {
            for (int pc : per_cu) {
                const int grid = pc * ncu;
                auto launch = [&](int c) {
                    if (shape == 0) {
                        if (u == 0) hipLaunchKernelGGL(read_flat<1>, dim3(grid), dim3(256), 0, 0, flat[c], flat_n, out);
                        else if (u == 1) hipLaunchKernelGGL(read_flat<2>, dim3(grid), dim3(256), 0, 0, flat[c], flat_n, out);
                        else hipLaunchKernelGGL(read_flat<4>, dim3(grid), dim3(256), 0, 0, flat[c], flat_n, out);
                    } else {
                        if (u == 0) hipLaunchKernelGGL(read_kick<1>, dim3(grid), dim3(256), 0, 0, velm[c], meta[c], force[c], N, N, out);
                        else if (u == 1) hipLaunchKernelGGL(read_kick<2>, dim3(grid), dim3(256), 0, 0, velm[c], meta[c], force[c], N, N, out);
                        else hipLaunchKernelGGL(read_kick<4>, dim3(grid), dim3(256), 0, 0, velm[c], meta[c], force[c], N, N, out);
                    }
                };
                for (int r = 0; r < 4; r++) launch(r % COPIES);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < REPS; r++) launch(r % COPIES);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / REPS;
                const double mb = shape == 0 ? flat_n * 32 / 1e6 : N * 56 / 1e6;
                printf("%s  loads in flight %d  work-groups/CU %d: %7.1f us  %5.2f TB/s%s\n", shape == 0 ? "flat" : "kick", 1 << u, pc, us,
                       mb / us, shape == 1 ? " (algorithmic; +7 % with the index word)" : "");
            }
        }
    }
    return 0;
}
