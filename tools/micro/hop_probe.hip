// hop_probe: round-trip latency of a flag hand-over between two work-groups of one launch that sit on different XCDs
// (work-group i runs on XCD i mod 8), for the memory kinds / scopes a device-side meeting can use.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hop_probe.hip -o build_variants/hop_probe && build_variants/hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SCOPE>
__global__ void pingpong(unsigned long long* a, unsigned long long* b, int n, int peer_block, unsigned long long* out) {
    // block 0 writes a, waits for b; block peer_block waits for a, writes b.  Other blocks idle (exit).
    if (threadIdx.x != 0) return;
    if (blockIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        for (int i = 1; i <= n; i++) {
            __hip_atomic_store(a, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
            int spin = 0;
            while (__hip_atomic_load(b, __ATOMIC_RELAXED, SCOPE) != (unsigned long long)i && ++spin < 10000000) {}
        }
        out[0] = wall_clock64() - t0;
    } else if ((int)blockIdx.x == peer_block) {
        for (int i = 1; i <= n; i++) {
            int spin = 0;
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, SCOPE) != (unsigned long long)i && ++spin < 10000000) {}
            __hip_atomic_store(b, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
        }
    }
}

// one-way fan-in: nblk blocks each store a tagged cell; block 0 polls all of them (like step_kernel's rows); time from block 0's
// own store to all seen, repeated n times with a return broadcast so the rounds stay in lockstep
template <int SCOPE>
__global__ void fanin(unsigned long long* rows, unsigned long long* bcast, int n, unsigned long long* out) {
    const int nb = gridDim.x;
    unsigned long long t_acc = 0;
    for (int i = 1; i <= n; i++) {
        if (threadIdx.x == 0) __hip_atomic_store(rows + blockIdx.x * 8, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
        if (blockIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            for (int r = threadIdx.x; r < nb; r += blockDim.x) {
                int spin = 0;
                while (__hip_atomic_load(rows + r * 8, __ATOMIC_RELAXED, SCOPE) != (unsigned long long)i && ++spin < 10000000) {}
            }
            __syncthreads();
            if (threadIdx.x == 0) { t_acc += wall_clock64() - t0; __hip_atomic_store(bcast, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE); }
        }
        if (threadIdx.x == 0) {
            int spin = 0;
            while (__hip_atomic_load(bcast, __ATOMIC_RELAXED, SCOPE) != (unsigned long long)i && ++spin < 10000000) {}
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t_acc;
}

int main() {
    unsigned long long *uc, *fg, *cg, *out;
    CK(hipExtMallocWithFlags((void**)&uc, 1 << 20, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&fg, 1 << 20, hipDeviceMallocFinegrained));
    CK(hipMalloc((void**)&cg, 1 << 20));
    CK(hipMalloc((void**)&out, 64));
    const int n = 2000;
    struct { const char* name; unsigned long long* p; } mem[3] = {{"uncached", uc}, {"fine-grained", fg}, {"coarse (hipMalloc)", cg}};
    for (int peer : {1, 8, 9}) {                 // XCD 1; XCD 0 (another CU of the same XCD); XCD 1
        for (auto& m : mem) {
            for (int scope = 0; scope < 2; scope++) {
                CK(hipMemset(m.p, 0, 1 << 20));
                unsigned long long h = 0;
                for (int rep = 0; rep < 2; rep++) {
                    if (scope == 0) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(16), dim3(64), 0, 0, m.p, m.p + 4096, n, peer, out);
                    else hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(16), dim3(64), 0, 0, m.p, m.p + 4096, n, peer, out);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
                    CK(hipMemset(m.p, 0, 1 << 20));
                }
                printf("pingpong  peer block %d  %-20s %-6s  round trip %.3f us (one way %.3f)\n", peer, m.name, scope ? "agent" : "system",
                       h / 100.0 / n, h / 200.0 / n);
            }
        }
    }
    for (auto& m : mem) {
        for (int scope = 0; scope < 2; scope++) {
            CK(hipMemset(m.p, 0, 1 << 20));
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; rep++) {
                if (scope == 0) hipLaunchKernelGGL(fanin<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(768), dim3(256), 0, 0, m.p, m.p + 65536, 500, out);
                else hipLaunchKernelGGL(fanin<__HIP_MEMORY_SCOPE_AGENT>, dim3(768), dim3(256), 0, 0, m.p, m.p + 65536, 500, out);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
                CK(hipMemset(m.p, 0, 1 << 20));
            }
            printf("fan-in 768 -> 1 (+ broadcast)  %-20s %-6s  collect %.3f us per round\n", m.name, scope ? "agent" : "system", h / 100.0 / 500);
        }
    }
    return 0;
}
