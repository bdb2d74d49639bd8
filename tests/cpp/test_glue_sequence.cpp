// GPU test of the OpenMM-HIP glue's call sequence (openmm_drudenose_amd/csrc/openmm_glue/HipDrudeTGNHKernels.cpp), which cannot
// be compiled here (no OpenMM): the same sequence -- per-step tgnh_set_*, tgnh_bind_buffers, plain pass structure, the
// fused path or the split path around the constraint / virtual-site / force call-outs, velocities changed behind the
// integrator's back between steps + stateChanged(), the status word read every step -- driven through
// include/DrudeTGNHIntegratorHip.hpp (DrudeTGNHIntegrator::execute) on device buffers in OpenMM's layouts, mixed precision.
// Inputs and outputs are flat binary files exchanged with tests/test_glue_sequence_gpu.py, which checks the results
// against tests/golden/oracle_regression.npz and against the oracle.
//
//   test_glue_sequence ints.bin doubles.bin out.bin
//   ints:    N P R G nclusters nsites nsteps perturb chains useDrudeChains useCOM precision flags | pairs[P][2] | resid[N] | group[N] |
//            cluster atoms[ncl][4] | site atoms[ns][4]
//   doubles: dt hardwall kDrude kTether tol | mass[N] | pos[N][3] | vel[N][3] | x0[N][3] | cluster dist[ncl][6] | site w[ns][3]
//   out:     pos[N][3] vel[N][3] etaDot[...]   (doubles)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "DrudeTGNHIntegratorHip.hpp"

#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(r_), __FILE__, __LINE__); return 2; } } while (0)
#define TG(e) do { if ((e) != TGNH_OK) { std::printf("tgnh error: %s (%s:%d)\n", tgnh_last_error(), __FILE__, __LINE__); return 3; } } while (0)

template <class T> static std::vector<T> slurp(const char* path) {
    std::vector<T> v;
    FILE* f = std::fopen(path, "rb");
    if (!f) return v;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize(n / sizeof(T));
    if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) v.clear();
    std::fclose(f);
    return v;
}

int main(int argc, char** argv) {
    using namespace drudetgnh;
    if (argc != 4) return 1;
    std::vector<int> I = slurp<int>(argv[1]);
    std::vector<double> D = slurp<double>(argv[2]);
    if (I.size() < 13 || D.size() < 5) { std::printf("bad input files\n"); return 1; }
    const int N = I[0], P = I[1], R = I[2], G = I[3], ncl = I[4], ns = I[5], nsteps = I[6], perturb = I[7];
    const int chains = I[8], useDrudeChains = I[9], useCOM = I[10], precision = I[11];    // TGNH_PREC_MIXED or _DOUBLE
    const bool dbl = precision == TGNH_PREC_DOUBLE;
    const int flags = I[12];                 // 0, TGNH_FLAG_RESIDENT_STEP (the glue's -DDRUDETGNH_RESIDENT_STEP build), TGNH_FLAG_TRUST_STATE_CHANGED (its -DDRUDETGNH_TRUST_STATE_CHANGED build), or both
    const int* pairs = &I[13]; const int* resid = pairs + 2 * P; const int* group = resid + N;
    const int* clAtoms = group + N; const int* siteAtoms = clAtoms + 4 * ncl;
    const double dt = D[0], hardwall = D[1], kDrude = D[2], kTether = D[3], tol = D[4];
    const double* mass = &D[5]; const double* pos0 = mass + N; const double* vel0 = pos0 + 3 * N; const double* x0h = vel0 + 3 * N;
    const double* clDist = x0h + 3 * N; const double* siteW = clDist + 6 * ncl;

    // ---- what Context construction does: System -> integrator -> kernel (the glue's initialize())
    SystemDesc sys;
    sys.mass.assign(mass, mass + N);
    for (int i = 0; i < P; i++) sys.drudePairs.push_back({pairs[2 * i], pairs[2 * i + 1]});
    sys.molecules.resize(R);
    for (int i = 0; i < N; i++) sys.molecules[resid[i]].push_back(i);
    static const int PAIR[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (int c = 0; c < ncl; c++)
        for (int k = 0; k < 6; k++)
            if (clDist[6 * c + k] > 0) sys.constraints.push_back({clAtoms[4 * c + PAIR[k][0]], clAtoms[4 * c + PAIR[k][1]]});
    DrudeTGNHIntegrator integ(300.0, 0.1, 1.0, 0.005, dt, 20, chains, useDrudeChains != 0, useCOM != 0);
    integ.setMaxDrudeDistance(hardwall);
    for (int g = 0; g < G; g++) integ.addTempGroup();
    for (int i = 0; i < N; i++) integ.addParticleTempGroup(group[i]);
    integ.initialize(sys, 0, TGNH_MODE_TGNH, precision, flags);            // the glue keeps the plain pass structure
    tgnh_handle h = integ.getHandle();
    const int padded = integ.getPaddedNumParticles();

    // ---- the platform's arrays, OpenMM layouts: mixed (float4 posq + float4 correction) or double (double4 posq)
    std::vector<float> posq(4 * N), corr(4 * N, 0.f), x0(4 * N);
    std::vector<double> posqd(4 * N), x0d(4 * N), velm(4 * N);
    std::vector<char> isDrude(N, 0);
    for (int i = 0; i < P; i++) isDrude[pairs[2 * i]] = 1;
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) {
            const double p = pos0[3 * i + k];
            posq[4 * i + k] = (float)p; corr[4 * i + k] = (float)(p - (double)posq[4 * i + k]);
            posqd[4 * i + k] = p;
            velm[4 * i + k] = vel0[3 * i + k];
            x0[4 * i + k] = (float)x0h[3 * i + k]; x0d[4 * i + k] = x0h[3 * i + k];
        }
        posq[4 * i + 3] = 0.f; posqd[4 * i + 3] = 0.0;
        velm[4 * i + 3] = mass[i] == 0.0 ? 0.0 : 1.0 / mass[i];
        const bool tether = mass[i] > 0 && !isDrude[i];                  // harness: tether every massive non-Drude site
        x0[4 * i + 3] = tether ? 1.f : 0.f; x0d[4 * i + 3] = tether ? 1.0 : 0.0;
    }
    const size_t rb = dbl ? 32 : 16;                                       // bytes of a real4
    void *d_posq, *d_corr, *d_velm, *d_force, *d_pd, *d_x0;
    HIPCHK(hipMalloc(&d_posq, rb * N)); HIPCHK(hipMalloc(&d_corr, 16 * N)); HIPCHK(hipMalloc(&d_velm, 32 * N));
    HIPCHK(hipMalloc(&d_force, 8 * 3 * (size_t)padded)); HIPCHK(hipMalloc(&d_pd, 32 * N)); HIPCHK(hipMalloc(&d_x0, rb * N));
    HIPCHK(hipMemcpy(d_posq, dbl ? (const void*)posqd.data() : (const void*)posq.data(), rb * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_corr, corr.data(), 16 * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_velm, velm.data(), 32 * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_x0, dbl ? (const void*)x0d.data() : (const void*)x0.data(), rb * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_force, 0, 8 * 3 * (size_t)padded)); HIPCHK(hipMemset(d_pd, 0, 32 * N));
    hipStream_t stream;
    HIPCHK(hipStreamCreate(&stream));
    DrudeTGNHIntegrator::Buffers buf{d_posq, dbl ? nullptr : d_corr, d_velm, d_force, d_pd};
    TG(tgnh_bind_buffers(h, buf.posq, buf.posqCorrection, buf.velm, buf.force, buf.posDelta));
    const bool constrained = ncl > 0 || ns > 0;
    if (ncl > 0) TG(tgnh_harness_set_clusters(h, ncl, clAtoms, clDist));
    if (ns > 0) TG(tgnh_harness_set_virtual_sites(h, ns, siteAtoms, siteW));

    // ---- the call-outs (stand-ins for OpenMM's: the library's harness kernels)
    int rc_callout = 0;
    DrudeTGNHIntegrator::CallOuts co;
    co.applyConstraints = [&] { if (ncl > 0) rc_callout |= tgnh_harness_shake_positions(h, tol, stream); };
    co.computeVirtualSites = [&] { if (ns > 0) rc_callout |= tgnh_harness_virtual_sites(h, stream); };
    co.calcForcesAndEnergy = [&] { rc_callout |= tgnh_harness_force(h, d_x0, kDrude, kTether, d_force, stream); };
    co.applyVelocityConstraints = [&] { if (ncl > 0) rc_callout |= tgnh_harness_shake_velocities(h, tol, stream); };
    co.calcForcesAndEnergy();                                          // Context::setPositions -> forces valid

    for (int step = 0; step < nsteps; step++) {
        if (perturb) {
            // Context::updateContextState (DrudeTGNHIntegrator.cpp:186): a CMMotionRemover takes the centre-of-mass velocity
            // out of every massive particle behind the integrator's back -> stateChanged() (.cpp:166-170)
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpy(velm.data(), d_velm, 32 * N, hipMemcpyDeviceToHost));
            double p[3] = {0, 0, 0}, m = 0;
            for (int i = 0; i < N; i++) if (mass[i] > 0) { for (int k = 0; k < 3; k++) p[k] += mass[i] * velm[4 * i + k]; m += mass[i]; }
            for (int i = 0; i < N; i++) if (mass[i] > 0) for (int k = 0; k < 3; k++) velm[4 * i + k] -= p[k] / m;
            HIPCHK(hipMemcpy(d_velm, velm.data(), 32 * N, hipMemcpyHostToDevice));
            integ.stateChanged();
        }
        try {
            integ.execute(stream, buf, co, constrained);
            const uint32_t flags = integ.checkStatus(stream);
            if (flags & ~1u) { std::printf("status word %u at step %d\n", flags, step); return 4; }
        } catch (const std::exception& e) { std::printf("exception at step %d: %s\n", step, e.what()); return 5; }
        if (rc_callout) { std::printf("call-out failed at step %d: %s\n", step, tgnh_last_error()); return 6; }
    }
    const double ke = integ.computeKineticEnergy(stream);              // cached sum of the last half step (Cu :654-658)
    double t = 0; int64_t count = 0;
    TG(tgnh_get_time(h, &t, &count));
    if (count != nsteps || std::fabs(t - nsteps * dt) > 1e-12) { std::printf("clock %g / %lld\n", t, (long long)count); return 7; }

    // ---- results
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(dbl ? (void*)posqd.data() : (void*)posq.data(), d_posq, rb * N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(corr.data(), d_corr, 16 * N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(velm.data(), d_velm, 32 * N, hipMemcpyDeviceToHost));
    int nEtaDot = 0;
    TG(tgnh_get_thermostat_len(h, 1, &nEtaDot));
    std::vector<double> etaDot(nEtaDot), out;
    TG(tgnh_get_thermostat_state(h, 1, stream, etaDot.data()));
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) out.push_back(dbl ? posqd[4 * i + k] : (double)posq[4 * i + k] + (double)corr[4 * i + k]);
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) out.push_back(velm[4 * i + k]);
    out.insert(out.end(), etaDot.begin(), etaDot.end());
    out.push_back(ke);
    FILE* f = std::fopen(argv[3], "wb");
    if (!f || std::fwrite(out.data(), sizeof(double), out.size(), f) != out.size()) return 8;
    std::fclose(f);
    integ.cleanup();
    std::printf("OK %d steps, %d slots, kinetic energy %.6f\n", nsteps, N, ke);
    return 0;
}
