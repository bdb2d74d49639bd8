// GPU test of the OpenMM-HIP glue ITSELF: the real sources of openmm_drudenose_amd/csrc/openmm_glue -- platforms/hip/src/
// HipDrudeTGNHKernels.cpp and serialization/src/*.cpp, unmodified -- compiled against tests/cpp/openmm_shim in its functional form
// (a miniature runtime behind OpenMM's signatures, NOT OpenMM: shim_runtime.cpp) and linked with libdrudetgnh_hip.so.  What runs
// here line by line is what an OpenMM build would run: registerKernelFactories() -> the factory -> createKernelImpl -> initialize()
// (System / DrudeForce / integrator -> tgnh_desc) -> execute() per step (per-step setters, tgnh_bind_buffers, the fused or the
// split sequence around the call-outs, the clock) -> computeKineticEnergy(); and, with a step count to split at, the XML proxy:
// serialize the integrator with its kernel's thermostat (DrudeTGNHThermostatStore), deserialize into a new integrator, a new
// kernel takes the parked state at initialize(), the run continues -- bit for bit what the uninterrupted run gives.
// The call-outs OpenMM would make (calcForcesAndEnergy, applyConstraints, applyVelocityConstraints, computeVirtualSites) are the
// library's harness kernels on a second handle bound to the same arrays.  Same file formats as tests/cpp/test_glue_sequence.cpp
// (the mirror of this sequence), one more argument:
//
//   test_glue_linked ints.bin doubles.bin out.bin [split_at]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "openmm/DrudeTGNHKernels.h"
#include "openmm/DrudeTGNHIntegrator.h"
#include "openmm/hip/HipPlatform.h"
#include "openmm/hip/HipContext.h"
#include "openmm/internal/ContextImpl.h"
#include "openmm/serialization/SerializationNode.h"
#include "openmm/serialization/DrudeTGNHIntegratorProxy.h"
#include "drude_tgnh.h"

extern "C" void registerKernelFactories();                      // the glue's plugin entry point
namespace OpenMM { const SerializationProxy* shimFindProxy(const std::type_info& type); }
using namespace OpenMM;

#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(r_), __FILE__, __LINE__); return 2; } } while (0)
#define TG(e) do { if ((e) != TGNH_OK) { std::printf("tgnh error: %s (%s:%d)\n", tgnh_last_error(), __FILE__, __LINE__); return 3; } } while (0)

// the node tree as OpenMM's XmlSerializer would write it: <Name prop="value" ...> children </Name>, the root with the proxy's type name
static void dumpXml(FILE* f, const SerializationNode& n, const std::string& name, const std::string& type, int depth) {
    std::fprintf(f, "%*s<%s", 2 * depth, "", name.c_str());
    if (!type.empty()) std::fprintf(f, " type=\"%s\"", type.c_str());
    for (const auto& p : n.shimProperties) std::fprintf(f, " %s=\"%s\"", p.first.c_str(), p.second.c_str());
    if (n.getChildren().empty()) { std::fprintf(f, "/>\n"); return; }
    std::fprintf(f, ">\n");
    for (const SerializationNode& c : n.getChildren()) dumpXml(f, c, c.getName(), "", depth + 1);
    std::fprintf(f, "%*s</%s>\n", 2 * depth, "", name.c_str());
}

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
    if (argc != 4 && argc != 5) return 1;
    std::vector<int> I = slurp<int>(argv[1]);
    std::vector<double> D = slurp<double>(argv[2]);
    const int splitAt = argc == 5 ? std::atoi(argv[4]) : 0;
    if (I.size() < 13 || D.size() < 5) { std::printf("bad input files\n"); return 1; }
    const int N = I[0], P = I[1], R = I[2], G = I[3], ncl = I[4], ns = I[5], nsteps = I[6], perturb = I[7];
    const int chains = I[8], useDrudeChains = I[9], useCOM = I[10], precision = I[11];
    const bool dbl = precision == TGNH_PREC_DOUBLE;
    const int* pairs = &I[13]; const int* resid = pairs + 2 * P; const int* group = resid + N;
    const int* clAtoms = group + N; const int* siteAtoms = clAtoms + 4 * ncl;
    const double dt = D[0], hardwall = D[1], kDrude = D[2], kTether = D[3], tol = D[4];
    const double* mass = &D[5]; const double* pos0 = mass + N; const double* vel0 = pos0 + 3 * N; const double* x0h = vel0 + 3 * N;
    const double* clDist = x0h + 3 * N; const double* siteW = clDist + 6 * ncl;
    static const int PAIR[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

    // ---- the objects a user builds: System, DrudeForce, (CMMotionRemover,) integrator
    System sys;
    sys.shimMass.assign(mass, mass + N);
    for (int c = 0; c < ncl; c++)
        for (int k = 0; k < 6; k++)
            if (clDist[6 * c + k] > 0) sys.shimConstraints.push_back({clAtoms[4 * c + PAIR[k][0]], clAtoms[4 * c + PAIR[k][1]], clDist[6 * c + k]});
    DrudeForce drude;
    for (int i = 0; i < P; i++) { drude.shimDrude.push_back(pairs[2 * i]); drude.shimParent.push_back(pairs[2 * i + 1]); }
    CMMotionRemover cmm;
    sys.shimForces.push_back(&drude);
    if (perturb == 1) sys.shimForces.push_back(&cmm);               // (perturb == 2: the same change of the velocities, made by a USER between steps -- Context::setVelocities -> stateChanged -- with no such Force in the System)
    std::unique_ptr<DrudeTGNHIntegrator> integ(new DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, dt, 20, chains, useDrudeChains != 0, useCOM != 0));
    integ->setMaxDrudeDistance(hardwall);
    integ->setConstraintTolerance(tol);
    for (int g = 0; g < G; g++) integ->addTempGroup();
    for (int i = 0; i < N; i++) integ->addParticleTempGroup(group[i]);
    auto molecules = [&](DrudeTGNHIntegrator& it) {                   // what DrudeTGNHIntegrator::initialize reads from Context::getMolecules (API :136-153)
        it.shimResId.assign(resid, resid + N);
        it.shimNumResidues = R;
    };
    molecules(*integ);

    // ---- the platform: arrays in OpenMM's layouts, the context, the call-outs
    const int padded = (N + 31) / 32 * 32;
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
        const bool tether = mass[i] > 0 && !isDrude[i];
        x0[4 * i + 3] = tether ? 1.f : 0.f; x0d[4 * i + 3] = tether ? 1.0 : 0.0;
    }
    const size_t rb = dbl ? 32 : 16;
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

    HipPlatform* platform = new HipPlatform();
    Platform::registerPlatform(platform);
    registerKernelFactories();                                         // the glue registers its factory with the "HIP" platform
    if (!platform->shimFactories.count(IntegrateDrudeTGNHStepKernel::Name())) { std::printf("no factory registered\n"); return 4; }
    const KernelFactory* factory = platform->shimFactories[IntegrateDrudeTGNHStepKernel::Name()];
    HipPlatform::PlatformData pdata;
    HipContext cu;
    cu.shimData = &pdata; pdata.contexts.push_back(&cu);
    cu.shimDouble = dbl; cu.shimMixed = !dbl; cu.shimPadded = padded; cu.shimStream = stream;
    cu.shimPosq.shimPtr = d_posq; cu.shimPosqCorrection.shimPtr = d_corr; cu.shimVelm.shimPtr = d_velm; cu.shimForce.shimPtr = d_force;
    cu.shimUtilities.shimPosDelta.shimPtr = d_pd;
    ContextImpl context;
    context.shimPlatformData = &pdata;

    // the call-outs: the library's harness kernels on a handle of their own, bound to the same arrays
    tgnh_handle hh = nullptr;
    {
        std::vector<int> pd(P), pp(P), ci, cj;
        for (int i = 0; i < P; i++) { pd[i] = pairs[2 * i]; pp[i] = pairs[2 * i + 1]; }
        tgnh_desc d = {};
        d.struct_size = sizeof(tgnh_desc);
        d.mode = TGNH_MODE_TGNH; d.precision = precision; d.device = 0;
        d.num_particles = N; d.padded_num_particles = padded; d.num_pairs = P; d.num_groups = G; d.num_residues = R;
        d.mass = mass; d.pair_drude = pd.data(); d.pair_parent = pp.data(); d.group = group; d.resid = resid;
        d.kB = BOLTZ; d.temperature = 300; d.coupling_time = 0.1; d.drude_temperature = 1; d.drude_coupling_time = 0.005;
        d.step_size = dt; d.drude_steps_per_real_step = 20; d.num_nh_chains = 1; d.use_drude_nh_chains = 1; d.use_com_temp_group = useCOM;
        TG(tgnh_create(&d, &hh));
        TG(tgnh_bind_buffers(hh, d_posq, dbl ? nullptr : d_corr, d_velm, d_force, d_pd));
        if (ncl > 0) TG(tgnh_harness_set_clusters(hh, ncl, clAtoms, clDist));
        if (ns > 0) TG(tgnh_harness_set_virtual_sites(hh, ns, siteAtoms, siteW));
    }
    int rc_callout = 0;
    context.shimForces = [&] { rc_callout |= tgnh_harness_force(hh, d_x0, kDrude, kTether, d_force, stream); };
    cu.shimUtilities.shimApplyConstraints = [&](double t) { if (ncl > 0) rc_callout |= tgnh_harness_shake_positions(hh, t, stream); };
    cu.shimUtilities.shimApplyVelocityConstraints = [&](double t) { if (ncl > 0) rc_callout |= tgnh_harness_shake_velocities(hh, t, stream); };
    cu.shimUtilities.shimVirtualSites = [&] { if (ns > 0) rc_callout |= tgnh_harness_virtual_sites(hh, stream); };

    // ---- Context construction: the kernel (through the factory the glue registered), initialize()
    std::unique_ptr<IntegrateDrudeTGNHStepKernel> kernel;
    auto make_kernel = [&]() -> int {
        try {
            KernelImpl* k = factory->createKernelImpl(IntegrateDrudeTGNHStepKernel::Name(), *platform, context);
            kernel.reset(dynamic_cast<IntegrateDrudeTGNHStepKernel*>(k));
            if (!kernel) { std::printf("the factory made something else\n"); return 5; }
            kernel->initialize(sys, *integ, drude);
        } catch (const std::exception& e) { std::printf("exception at initialize: %s\n", e.what()); return 5; }
        return 0;
    };
    if (int rc = make_kernel()) return rc;
    context.calcForcesAndEnergy(true, false);                          // Context::setPositions -> forces valid

    for (int step = 0; step < nsteps; step++) {
        if (splitAt > 0 && step == splitAt) {
            // checkpoint: the integrator through its XML proxy (found in the registry the library's constructor filled), thermostat
            // and clock from the live kernel; a new integrator out of the node; the old kernel and integrator go; a new kernel
            const SerializationProxy* proxy = shimFindProxy(typeid(DrudeTGNHIntegrator));
            if (!proxy) { std::printf("no proxy registered\n"); return 9; }
            SerializationNode node;
            try {
                proxy->serialize(integ.get(), node);
                bool has = false;
                for (const SerializationNode& c : node.getChildren()) has = has || c.getName() == "ThermostatState";
                if (!has) { std::printf("no ThermostatState in the node\n"); return 9; }
                if (FILE* xf = std::fopen((std::string(argv[3]) + ".xml").c_str(), "w")) {       // for the Python reader (openmm_drudenose_amd/serialization.py)
                    dumpXml(xf, node, "Integrator", proxy->shimTypeName, 0);
                    std::fclose(xf);
                }
                DrudeTGNHIntegrator* fresh = reinterpret_cast<DrudeTGNHIntegrator*>(proxy->deserialize(node));
                kernel.reset();
                integ.reset(fresh);
            } catch (const std::exception& e) { std::printf("exception in the proxy: %s\n", e.what()); return 9; }
            molecules(*integ);
            if (integ->getNumTempGroups() != G || integ->getMaxDrudeDistance() != hardwall || integ->getConstraintTolerance() != tol ||
                integ->getStepSize() != dt || integ->getNumNHChains() != chains) { std::printf("the proxy lost a property\n"); return 9; }
            if (int rc = make_kernel()) return rc;
        }
        if (perturb) {
            // Context::updateContextState (API :186): a CMMotionRemover takes the centre-of-mass velocity out behind the
            // integrator's back -> stateChanged() (API :166-170)
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpy(velm.data(), d_velm, 32 * N, hipMemcpyDeviceToHost));
            double p[3] = {0, 0, 0}, m = 0;
            for (int i = 0; i < N; i++) if (mass[i] > 0) { for (int k = 0; k < 3; k++) p[k] += mass[i] * velm[4 * i + k]; m += mass[i]; }
            for (int i = 0; i < N; i++) if (mass[i] > 0) for (int k = 0; k < 3; k++) velm[4 * i + k] -= p[k] / m;
            if (perturb == 2)                                          // the USER's change: every velocity 3 % smaller (6 % of every kinetic energy: a stale sum shows)
                for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) velm[4 * i + k] *= 0.97;
            HIPCHK(hipMemcpy(d_velm, velm.data(), 32 * N, hipMemcpyHostToDevice));
            integ->shimKEValid = false;
        }
        try {
            kernel->execute(context, *integ);
            integ->shimKEValid = true;                                 // API :192
        } catch (const std::exception& e) { std::printf("exception at step %d: %s\n", step, e.what()); return 5; }
        if (rc_callout) { std::printf("call-out failed at step %d: %s\n", step, tgnh_last_error()); return 6; }
    }
    double ke = 0;
    try { ke = kernel->computeKineticEnergy(context, *integ, true); }          // cached sum of the last half step (Cu :654-658)
    catch (const std::exception& e) { std::printf("exception in computeKineticEnergy: %s\n", e.what()); return 5; }
    if (cu.shimStepCount != nsteps || std::fabs(cu.shimTime - nsteps * dt) > 1e-12 || cu.shimReorders != nsteps) {
        std::printf("clock %g / %lld / %d\n", cu.shimTime, cu.shimStepCount, cu.shimReorders); return 7;
    }

    // ---- results (the thermostat through the reader the kernel registered)
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(dbl ? (void*)posqd.data() : (void*)posq.data(), d_posq, rb * N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(corr.data(), d_corr, 16 * N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(velm.data(), d_velm, 32 * N, hipMemcpyDeviceToHost));
    DrudeTGNHThermostatState state;
    if (!DrudeTGNHThermostatStore::read(integ.get(), state)) { std::printf("no thermostat reader\n"); return 8; }
    if (state.stepCount != nsteps) { std::printf("the library's clock says %lld steps\n", state.stepCount); return 7; }
    std::vector<double> out;
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) out.push_back(dbl ? posqd[4 * i + k] : (double)posq[4 * i + k] + (double)corr[4 * i + k]);
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) out.push_back(velm[4 * i + k]);
    out.insert(out.end(), state.etaDot.begin(), state.etaDot.end());
    out.push_back(ke);
    FILE* f = std::fopen(argv[3], "wb");
    if (!f || std::fwrite(out.data(), sizeof(double), out.size(), f) != out.size()) return 8;
    std::fclose(f);
    kernel.reset();
    tgnh_destroy(hh);
    std::printf("OK %d steps, %d slots, kinetic energy %.6f, %d force call-outs\n", nsteps, N, ke, context.shimForceCalls);
    return 0;
}
