// tgnh_harness.hip -- harness call-outs of the constrained path: SHAKE on posDelta, the velocity stage,
// three-particle-average virtual sites.  NOT part of the reference's plugin: it delegates all three to OpenMM
// (Cu :363 applyConstraints, :391 applyVelocityConstraints, :377 computeVirtualSites; Ref :268, :373).
// They exist so that the split step path (tgnh_step_begin_kick / _move / _end_kick / _end_thermo) can be
// exercised and measured with real constraint corrections, the way the harness force stands in for
// calcForcesAndEnergy.  The oracle carries the same algorithms in fp64 (oracle/tgnh_oracle.c).
//
// Constraint cluster: up to 4 atoms and the 6 possible distances among them in the fixed order
// (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); distance 0 = no constraint.  One lane per cluster, SHAKE sweeps in that
// order until every |d^2 - r^2| <= 2 tol d^2 -- everything in registers, static indices.
#include "tgnh_internal.h"

namespace tgnh {

template <int PREC> struct HPrec;
template <> struct HPrec<TGNH_PREC_SINGLE> { typedef float real; typedef float mixed; typedef float4 real4; typedef float4 mixed4; };
template <> struct HPrec<TGNH_PREC_MIXED>  { typedef float real; typedef double mixed; typedef float4 real4; typedef double4 mixed4; };
template <> struct HPrec<TGNH_PREC_DOUBLE> { typedef double real; typedef double mixed; typedef double4 real4; typedef double4 mixed4; };

struct ClusterArgs {
    const int4* atoms;        // [n] slot indices, -1 = unused
    const double* dist;       // [n][6]
    int n;
    const void* posq; const void* posq_corr;
    void* velm; void* pos_delta;
    double tol;
    uint32_t* status;         // bit1: SHAKE did not converge
};

constexpr int SHAKE_MAX_ITER = 500;
__device__ constexpr int PA[6] = {0, 0, 0, 1, 1, 2};
__device__ constexpr int PB[6] = {1, 2, 3, 2, 3, 3};

template <int PREC, bool VELOCITY>
__global__ __launch_bounds__(BLOCK) void shake_kernel(const ClusterArgs a) {
    typedef typename HPrec<PREC>::real4 real4;
    typedef typename HPrec<PREC>::mixed mixed;
    typedef typename HPrec<PREC>::mixed4 mixed4;
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    mixed4* __restrict__ velm = reinterpret_cast<mixed4*>(a.velm);
    mixed4* __restrict__ pdelta = reinterpret_cast<mixed4*>(a.pos_delta);
    // The iteration runs in double in every precision: in single, a cluster's positions (~3 nm, ulp 2.4e-7) put the squared
    // bond lengths within a few roundings of the 2 tol d^2 = 2e-7 nm^2 gate, and a float iteration can circle it for good
    // (status bit 1 in one of 48 trajectories of the reference's water test, round 4) -- a property of this call-out, not of the path
    typedef double work;
    const work tol = (work)a.tol;
    for (int c = blockIdx.x * BLOCK + threadIdx.x; c < a.n; c += gridDim.x * BLOCK) {
        const int4 at4 = a.atoms[c];
        const int at[4] = {at4.x, at4.y, at4.z, at4.w};
        work x[4][3], q[4][3], w[4];          // positions, the corrected quantity (delta or velocity), inverse masses
        mixed4 keep[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w[k] = 0;
#pragma unroll
            for (int j = 0; j < 3; j++) { x[k][j] = 0; q[k][j] = 0; }
            if (at[k] >= 0) {
                const real4 p = posq[at[k]];
                x[k][0] = p.x; x[k][1] = p.y; x[k][2] = p.z;
                if (PREC == TGNH_PREC_MIXED) { const float4 cc = pcorr[at[k]]; x[k][0] += (work)cc.x; x[k][1] += (work)cc.y; x[k][2] += (work)cc.z; }
                const mixed4 v = velm[at[k]];
                w[k] = v.w;
                if (VELOCITY) { keep[k] = v; q[k][0] = v.x; q[k][1] = v.y; q[k][2] = v.z; }
                else { keep[k] = pdelta[at[k]]; q[k][0] = keep[k].x; q[k][1] = keep[k].y; q[k][2] = keep[k].z; }
            }
        }
        work d2[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { const work d = (work)a.dist[(size_t)c * 6 + k]; d2[k] = d * d; }
        bool converged = false;
        int iter = 0;
        while (!converged && iter++ < SHAKE_MAX_ITER) {
            converged = true;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (d2[k] > 0) {
                    const int i = PA[k], j = PB[k];
                    const work rx = x[i][0] - x[j][0], ry = x[i][1] - x[j][1], rz = x[i][2] - x[j][2];
                    if (VELOCITY) {
                        const work vx = q[i][0] - q[j][0], vy = q[i][1] - q[j][1], vz = q[i][2] - q[j][2];
                        const work r2 = rx * rx + ry * ry + rz * rz, rv = rx * vx + ry * vy + rz * vz;
                        const work g = rv / (r2 * (w[i] + w[j]));
                        if ((g < 0 ? -g : g) * (w[i] + w[j]) > tol) {
                            converged = false;
                            q[i][0] -= g * w[i] * rx; q[i][1] -= g * w[i] * ry; q[i][2] -= g * w[i] * rz;
                            q[j][0] += g * w[j] * rx; q[j][1] += g * w[j] * ry; q[j][2] += g * w[j] * rz;
                        }
                    } else {
                        const work sx = rx + (q[i][0] - q[j][0]), sy = ry + (q[i][1] - q[j][1]), sz = rz + (q[i][2] - q[j][2]);
                        const work diff = d2[k] - (sx * sx + sy * sy + sz * sz);
                        if ((diff < 0 ? -diff : diff) > 2 * tol * d2[k]) {
                            converged = false;
                            const work g = diff / (2 * (rx * sx + ry * sy + rz * sz) * (w[i] + w[j]));
                            q[i][0] += g * w[i] * rx; q[i][1] += g * w[i] * ry; q[i][2] += g * w[i] * rz;
                            q[j][0] -= g * w[j] * rx; q[j][1] -= g * w[j] * ry; q[j][2] -= g * w[j] * rz;
                        }
                    }
                }
            }
        }
        if (!converged) atomicOr(a.status, 2u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (at[k] >= 0) {
                mixed4 o = keep[k];
                o.x = (mixed)q[k][0]; o.y = (mixed)q[k][1]; o.z = (mixed)q[k][2];
                if (VELOCITY) velm[at[k]] = o; else pdelta[at[k]] = o;
            }
        }
    }
}

struct SiteArgs {
    const int4* atoms;        // [n] (site, p1, p2, p3)
    const double* w;          // [n][3]
    int n;
    void* posq; void* posq_corr;
};

template <int PREC>
__global__ __launch_bounds__(BLOCK) void site_kernel(const SiteArgs a) {
    typedef typename HPrec<PREC>::real real;
    typedef typename HPrec<PREC>::real4 real4;
    typedef typename HPrec<PREC>::mixed mixed;
    real4* __restrict__ posq = reinterpret_cast<real4*>(a.posq);
    float4* __restrict__ pcorr = reinterpret_cast<float4*>(a.posq_corr);
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < a.n; k += gridDim.x * BLOCK) {
        const int4 at = a.atoms[k];
        const int src[3] = {at.y, at.z, at.w};
        mixed s[3] = {0, 0, 0};
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const real4 p = posq[src[m]];
            mixed x = p.x, y = p.y, z = p.z;
            if (PREC == TGNH_PREC_MIXED) { const float4 c = pcorr[src[m]]; x += (mixed)c.x; y += (mixed)c.y; z += (mixed)c.z; }
            const mixed wm = (mixed)a.w[(size_t)k * 3 + m];
            s[0] += wm * x; s[1] += wm * y; s[2] += wm * z;
        }
        real4 o = posq[at.x];
        if (PREC == TGNH_PREC_MIXED) {
            const float hx = (float)s[0], hy = (float)s[1], hz = (float)s[2];
            o.x = hx; o.y = hy; o.z = hz;
            pcorr[at.x] = make_float4((float)(s[0] - hx), (float)(s[1] - hy), (float)(s[2] - hz), 0.0f);
        } else {
            o.x = (real)s[0]; o.y = (real)s[1]; o.z = (real)s[2];
        }
        posq[at.x] = o;
    }
}

static int grid_of(int n) { int g = (n + BLOCK - 1) / BLOCK; return g < 1 ? 1 : (g > 2048 ? 2048 : g); }

template <bool VEL>
static hipError_t launch_shake(int precision, const ClusterArgs& a, hipStream_t s) {
    const int g = grid_of(a.n);
    switch (precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((shake_kernel<TGNH_PREC_SINGLE, VEL>), dim3(g), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((shake_kernel<TGNH_PREC_MIXED, VEL>), dim3(g), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_DOUBLE: TGNH_LAUNCH((shake_kernel<TGNH_PREC_DOUBLE, VEL>), dim3(g), dim3(BLOCK), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static hipError_t launch_sites(int precision, const SiteArgs& a, hipStream_t s) {
    const int g = grid_of(a.n);
    switch (precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((site_kernel<TGNH_PREC_SINGLE>), dim3(g), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((site_kernel<TGNH_PREC_MIXED>), dim3(g), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_DOUBLE: TGNH_LAUNCH((site_kernel<TGNH_PREC_DOUBLE>), dim3(g), dim3(BLOCK), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace tgnh

namespace tgnh {
// ---------------------------------------------------------------------------
// The force field and the CMMotionRemover of the reference's testWater (TestReferenceDrudeTGNHIntegrator.cpp:111-166), as
// harness call-outs: what that test asks OpenMM for (reaction-field NonbondedForce, cutoff 1 nm, + DrudeForce + the M site's
// force spread over O, H1, H2; oracle/water_ff.c is the CPU statement the tests check this against).  SWM4-NDP layout
// O, D, H1, H2, M per molecule.  One work-group per molecule: its threads walk the other molecules (25 site pairs each, all
// fp64), the 15 force components of the molecule's own sites are reduced over the work-group in a fixed order.
// ---------------------------------------------------------------------------
struct WaterArgs {
    const void* posq; const void* posq_corr;
    long long* force;
    int n_mol, padded;
    double box, cutoff;
};

__device__ __forceinline__ double h_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int PREC>
__global__ __launch_bounds__(BLOCK) void water_force_kernel(const WaterArgs a) {
    typedef typename HPrec<PREC>::real4 real4;
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    __shared__ double sred[BLOCK / 64][15];
    const double Q[5] = {1.71636, -1.71636, 0.55733, 0.55733, -1.11466};
    const double W[3] = {0.786646558, 0.106676721, 0.106676721};
    const double ONE_4PI_EPS0 = 138.935456, eps_rf = 78.3;
    const double krf = (1.0 / (a.cutoff * a.cutoff * a.cutoff)) * (eps_rf - 1.0) / (2.0 * eps_rf + 1.0);
    const double sigma = 0.318395, eps = 0.21094 * 4.184, kd = 100000.0 * 4.184;
    const double c2 = a.cutoff * a.cutoff, inv_box = 1.0 / a.box;
    auto site = [&](int idx, double* x) {
        const real4 p = posq[idx];
        x[0] = p.x; x[1] = p.y; x[2] = p.z;
        if (PREC == TGNH_PREC_MIXED) { const float4 c = pcorr[idx]; x[0] += (double)c.x; x[1] += (double)c.y; x[2] += (double)c.z; }
    };
    const int ma = blockIdx.x;
    double xa[5][3], f[5][3];
#pragma unroll
    for (int i = 0; i < 5; i++) { site(5 * ma + i, xa[i]); f[i][0] = f[i][1] = f[i][2] = 0.0; }
    for (int mb = threadIdx.x; mb < a.n_mol; mb += BLOCK) {
        if (mb == ma) continue;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            double xb[3];
            site(5 * mb + j, xb);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                double d[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { d[k] = xa[i][k] - xb[k]; d[k] -= a.box * floor(d[k] * inv_box + 0.5); }
                const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                if (r2 >= c2) continue;
                const double r = sqrt(r2), inv_r = 1.0 / r;
                double dEdr = ONE_4PI_EPS0 * Q[i] * Q[j] * (-inv_r * inv_r + 2.0 * krf * r);
                if (i == 0 && j == 0) {
                    const double s2 = sigma * sigma / r2, s6 = s2 * s2 * s2;
                    dEdr += 4.0 * eps * (-12.0 * s6 * s6 + 6.0 * s6) * inv_r;
                }
                const double c = -dEdr * inv_r;
                f[i][0] += c * d[0]; f[i][1] += c * d[1]; f[i][2] += c * d[2];
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double t = h_wave_sum(f[i][k]);
            if (lane == 0) sred[wv][3 * i + k] = t;
        }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        double fs[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { fs[i] = 0.0; for (int w = 0; w < BLOCK / 64; w++) fs[i] += sred[w][3 * i + k]; }
        const double sp = kd * (xa[1][k] - xa[0][k]);                   // Drude spring (DrudeForce, test :148)
        fs[1] -= sp; fs[0] += sp;
        fs[0] += W[0] * fs[4]; fs[2] += W[1] * fs[4]; fs[3] += W[2] * fs[4]; fs[4] = 0.0;   // the M site's force (test :147)
#pragma unroll
        for (int i = 0; i < 5; i++) a.force[(size_t)k * a.padded + 5 * ma + i] = (long long)(fs[i] * 4294967296.0);
    }
}

// OpenMM's CMMotionRemover (frequency 1): subtract the centre-of-mass velocity from every massive particle.  One work-group.
template <int PREC>
__global__ __launch_bounds__(BLOCK) void cmm_kernel(void* velm_, int n) {
    typedef typename HPrec<PREC>::mixed mixed;
    typedef typename HPrec<PREC>::mixed4 mixed4;
    mixed4* __restrict__ velm = reinterpret_cast<mixed4*>(velm_);
    __shared__ double sred[BLOCK / 64][4];
    __shared__ double vcm[3];
    double px = 0, py = 0, pz = 0, pm = 0;
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        const mixed4 v = velm[i];
        if (v.w != 0) { const double m = 1.0 / (double)v.w; px += m * v.x; py += m * v.y; pz += m * v.z; pm += m; }
    }
    px = h_wave_sum(px); py = h_wave_sum(py); pz = h_wave_sum(pz); pm = h_wave_sum(pm);
    if ((threadIdx.x & 63) == 0) { double* r = sred[threadIdx.x >> 6]; r[0] = px; r[1] = py; r[2] = pz; r[3] = pm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0, 0, 0, 0};
        for (int w = 0; w < BLOCK / 64; w++) for (int k = 0; k < 4; k++) t[k] += sred[w][k];
        for (int k = 0; k < 3; k++) vcm[k] = t[k] / t[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        mixed4 v = velm[i];
        if (v.w != 0) { v.x -= (mixed)vcm[0]; v.y -= (mixed)vcm[1]; v.z -= (mixed)vcm[2]; velm[i] = v; }
    }
}

}  // namespace tgnh
using namespace tgnh;

#define H_FAIL(code, msg) do { tgnh_set_error(msg); return code; } while (0)
#define H_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { tgnh_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return TGNH_ERR_HIP; } } while (0)

static tgnh_status harness_ready(tgnh_handle h) {
    if (!h) H_FAIL(TGNH_ERR_ARG, "null handle");
    if (h->host_only) H_FAIL(TGNH_ERR_STATE, "host-only handle (device -1): no GPU work can be launched on it");
    if (!h->velm) H_FAIL(TGNH_ERR_STATE, "tgnh_bind_buffers has not been called");
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_harness_set_clusters(tgnh_handle h, int n, const int32_t* atoms, const double* dist) {
    if (!h) H_FAIL(TGNH_ERR_ARG, "null handle");
    if (h->host_only) H_FAIL(TGNH_ERR_STATE, "host-only handle");
    if (n < 0 || (n > 0 && (!atoms || !dist))) H_FAIL(TGNH_ERR_ARG, "bad cluster arrays");
    for (long long i = 0; i < 4LL * n; i++)
        if (atoms[i] < -1 || atoms[i] >= h->d.num_particles) H_FAIL(TGNH_ERR_ARG, "cluster atom index out of range");
    H_HIP(hipSetDevice(h->device));
    if (h->d_cl_atoms) { (void)hipFree(h->d_cl_atoms); h->d_cl_atoms = nullptr; }
    if (h->d_cl_dist) { (void)hipFree(h->d_cl_dist); h->d_cl_dist = nullptr; }
    h->num_clusters = n;
    if (n == 0) return TGNH_OK;
    H_HIP(hipMalloc(&h->d_cl_atoms, sizeof(int4) * n));
    H_HIP(hipMemcpy(h->d_cl_atoms, atoms, sizeof(int4) * n, hipMemcpyHostToDevice));
    H_HIP(hipMalloc(&h->d_cl_dist, sizeof(double) * 6 * n));
    H_HIP(hipMemcpy(h->d_cl_dist, dist, sizeof(double) * 6 * n, hipMemcpyHostToDevice));
    return TGNH_OK;
}

static ClusterArgs cluster_args(tgnh_handle h, double tol) {
    ClusterArgs a{};
    a.atoms = h->d_cl_atoms; a.dist = h->d_cl_dist; a.n = h->num_clusters;
    a.posq = h->posq; a.posq_corr = h->posq_corr; a.velm = h->velm; a.pos_delta = h->pos_delta;
    a.tol = tol; a.status = h->d_status;
    return a;
}

extern "C" tgnh_status tgnh_harness_shake_positions(tgnh_handle h, double tol, void* stream) {
    tgnh_status rc = harness_ready(h); if (rc) return rc;
    if (!h->pos_delta) H_FAIL(TGNH_ERR_STATE, "posDelta buffer not bound");
    if (h->num_clusters == 0) return TGNH_OK;
    H_HIP(hipSetDevice(h->device));
    H_HIP(launch_shake<false>(h->d.precision, cluster_args(h, tol), (hipStream_t)stream));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_harness_shake_velocities(tgnh_handle h, double tol, void* stream) {
    tgnh_status rc = harness_ready(h); if (rc) return rc;
    h->ke_carry = false;                 // velocities are about to change behind the integrator (TRUST_STATE_CHANGED: recompute)
    if (h->num_clusters == 0) return TGNH_OK;
    H_HIP(hipSetDevice(h->device));
    H_HIP(launch_shake<true>(h->d.precision, cluster_args(h, tol), (hipStream_t)stream));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_harness_set_virtual_sites(tgnh_handle h, int n, const int32_t* atoms, const double* weights) {
    if (!h) H_FAIL(TGNH_ERR_ARG, "null handle");
    if (h->host_only) H_FAIL(TGNH_ERR_STATE, "host-only handle");
    if (n < 0 || (n > 0 && (!atoms || !weights))) H_FAIL(TGNH_ERR_ARG, "bad virtual-site arrays");
    for (long long i = 0; i < 4LL * n; i++)
        if (atoms[i] < 0 || atoms[i] >= h->d.num_particles) H_FAIL(TGNH_ERR_ARG, "virtual-site atom index out of range");
    H_HIP(hipSetDevice(h->device));
    if (h->d_vs_atoms) { (void)hipFree(h->d_vs_atoms); h->d_vs_atoms = nullptr; }
    if (h->d_vs_w) { (void)hipFree(h->d_vs_w); h->d_vs_w = nullptr; }
    h->num_sites = n;
    if (n == 0) return TGNH_OK;
    H_HIP(hipMalloc(&h->d_vs_atoms, sizeof(int4) * n));
    H_HIP(hipMemcpy(h->d_vs_atoms, atoms, sizeof(int4) * n, hipMemcpyHostToDevice));
    H_HIP(hipMalloc(&h->d_vs_w, sizeof(double) * 3 * n));
    H_HIP(hipMemcpy(h->d_vs_w, weights, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_harness_virtual_sites(tgnh_handle h, void* stream) {
    tgnh_status rc = harness_ready(h); if (rc) return rc;
    if (h->num_sites == 0) return TGNH_OK;
    H_HIP(hipSetDevice(h->device));
    SiteArgs a{};
    a.atoms = h->d_vs_atoms; a.w = h->d_vs_w; a.n = h->num_sites; a.posq = h->posq; a.posq_corr = h->posq_corr;
    H_HIP(launch_sites(h->d.precision, a, (hipStream_t)stream));
    return TGNH_OK;
}

// nsteps x the constrained step (Cu :336-406) with the harness call-outs in OpenMM's places
extern "C" tgnh_status tgnh_run_harness_constrained(tgnh_handle h, const void* x0, double k_drude, double k_tether,
                                                    double tol, int nsteps, void* stream) {
    tgnh_status rc = harness_ready(h); if (rc) return rc;
    for (int i = 0; i < nsteps; i++) {
        rc = tgnh_step_begin_kick(h, stream); if (rc) return rc;                         // Cu :336-360
        rc = tgnh_harness_shake_positions(h, tol, stream); if (rc) return rc;            // Cu :363 call-out
        rc = tgnh_step_begin_move(h, stream); if (rc) return rc;                         // Cu :366-376
        rc = tgnh_harness_virtual_sites(h, stream); if (rc) return rc;                   // Cu :377 call-out
        rc = tgnh_harness_force(h, x0, k_drude, k_tether, const_cast<void*>(h->force), stream); if (rc) return rc;   // Cu :380 call-out
        rc = tgnh_step_end_kick(h, stream); if (rc) return rc;                           // Cu :384-388
        if (h->d.mode == TGNH_MODE_TGNH) {                                               // Cu :391 call-out (the Reference platform has none)
            rc = tgnh_harness_shake_velocities(h, tol, stream); if (rc) return rc;
        }
        rc = tgnh_step_end_thermo(h, stream); if (rc) return rc;                         // Cu :394-406
    }
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// testWater call-outs (water_force_kernel, cmm_kernel above)
// ---------------------------------------------------------------------------
extern "C" tgnh_status tgnh_harness_water_force(tgnh_handle h, double box, double cutoff, void* force_out, void* stream) {
    tgnh_status rc = harness_ready(h); if (rc) return rc;
    if (!force_out || !(box > 0) || !(cutoff > 0)) H_FAIL(TGNH_ERR_ARG, "bad box / cutoff / force buffer");
    if (h->d.num_particles % 5) H_FAIL(TGNH_ERR_ARG, "the testWater force field needs O, D, H1, H2, M per molecule");
    H_HIP(hipSetDevice(h->device));
    WaterArgs a{};
    a.posq = h->posq; a.posq_corr = h->posq_corr; a.force = reinterpret_cast<long long*>(force_out);
    a.n_mol = h->d.num_particles / 5; a.padded = h->d.padded_num_particles; a.box = box; a.cutoff = cutoff;
    hipStream_t s = (hipStream_t)stream;
    switch (h->d.precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((water_force_kernel<TGNH_PREC_SINGLE>), dim3(a.n_mol), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((water_force_kernel<TGNH_PREC_MIXED>), dim3(a.n_mol), dim3(BLOCK), 0, s, a); break;
        default: TGNH_LAUNCH((water_force_kernel<TGNH_PREC_DOUBLE>), dim3(a.n_mol), dim3(BLOCK), 0, s, a); break;
    }
    H_HIP(hipGetLastError());
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_harness_remove_cm_motion(tgnh_handle h, void* stream) {
    tgnh_status rc = harness_ready(h); if (rc) return rc;
    h->ke_carry = false;                 // (OpenMM's CMMotionRemover does not tell the integrator: the glue does not set TRUST_STATE_CHANGED beside one)
    H_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    switch (h->d.precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((cmm_kernel<TGNH_PREC_SINGLE>), dim3(1), dim3(BLOCK), 0, s, h->velm, h->d.num_particles); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((cmm_kernel<TGNH_PREC_MIXED>), dim3(1), dim3(BLOCK), 0, s, h->velm, h->d.num_particles); break;
        default: TGNH_LAUNCH((cmm_kernel<TGNH_PREC_DOUBLE>), dim3(1), dim3(BLOCK), 0, s, h->velm, h->d.num_particles); break;
    }
    H_HIP(hipGetLastError());
    return TGNH_OK;
}
