// issue_probe: cycles per fp64 instruction for ONE wavefront per SIMD, dependent chain vs independent streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int INDEP>
__global__ void k(double* out, unsigned long long* t, int iters) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    const double c = 0.999999, d = 1e-9;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (INDEP == 1) {              // 8 dependent fmas
#pragma unroll
            for (int i = 0; i < 8; i++) a[0] = fma(a[0], c, d);
        } else {                       // INDEP independent streams, 8 fmas in total
#pragma unroll
            for (int i = 0; i < 8; i++) a[i % INDEP] = fma(a[i % INDEP], c, d);
        }
    }
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) t[0] = c1 - c0;
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];
    if (s == 1234.5) out[0] = s;
}
int main() {
    double* out; unsigned long long* t; CK(hipMalloc((void**)&out, 64)); CK(hipMalloc((void**)&t, 64));
    const int iters = 20000; unsigned long long h;
#define RUN(N) for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, out, t, iters); CK(hipDeviceSynchronize()); } \
    CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); printf("%d independent stream(s): %.2f cycles per fp64 fma\n", N, (double)h / (8.0 * iters));
    RUN(1) RUN(2) RUN(4) RUN(8)
    return 0;
}
