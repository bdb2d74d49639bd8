// DrudeTGNHIntegratorHip.hpp -- C++ host-side mirror of the reference's API class over the C ABI.
//
// OpenMM is not needed: `SystemDesc` carries what DrudeTGNHIntegrator::initialize reads from an OpenMM System,
// DrudeForce and Context::getMolecules (openmmapi/src/DrudeTGNHIntegrator.cpp:103-160).  Method names, argument
// meaning, defaults and error behaviour follow openmmapi/include/openmm/DrudeTGNHIntegrator.h:56-311 of
// scychon/openmm_drudeNose; errors are std::runtime_error where the reference throws OpenMMException.
// Inside OpenMM use the plugin glue instead (openmm_drudenose_amd/csrc/openmm_glue).
#ifndef DRUDE_TGNH_INTEGRATOR_HIP_HPP_
#define DRUDE_TGNH_INTEGRATOR_HIP_HPP_

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "drude_tgnh.h"

namespace drudetgnh {

struct SystemDesc {
    std::vector<double> mass;                      // System::getParticleMass
    std::vector<std::pair<int, int>> drudePairs;   // DrudeForce::getParticleParameters (p, p1), DrudeForce order
    std::vector<std::pair<int, int>> constraints;  // System::getConstraintParameters
    std::vector<std::vector<int>> molecules;       // Context::getMolecules
    bool hasCMMotionRemover = false;
};

class DrudeTGNHIntegrator {
public:
    DrudeTGNHIntegrator(double temperature, double couplingTime, double drudeTemperature, double drudeCouplingTime,
                        double stepSize, int drudeStepsPerRealStep = 20, int numNHChains = 1,
                        bool useDrudeNHChains = false, bool useCOMTempGroup = true)
        : temperature(temperature), couplingTime(couplingTime), drudeTemperature(drudeTemperature),
          drudeCouplingTime(drudeCouplingTime), maxDrudeDistance(0), stepSize(stepSize), constraintTolerance(1e-5),
          drudeStepsPerRealStep(drudeStepsPerRealStep), numNHChains(numNHChains), useDrudeNHChains(useDrudeNHChains),
          useCOMTempGroup(useCOMTempGroup) {}
    ~DrudeTGNHIntegrator() { cleanup(); }
    DrudeTGNHIntegrator(const DrudeTGNHIntegrator&) = delete;
    DrudeTGNHIntegrator& operator=(const DrudeTGNHIntegrator&) = delete;

    double getTemperature() const { return temperature; }
    void setTemperature(double t) { temperature = t; }
    double getCouplingTime() const { return couplingTime; }
    void setCouplingTime(double tau) { couplingTime = tau; }
    double getDrudeTemperature() const { return drudeTemperature; }
    void setDrudeTemperature(double t) { drudeTemperature = t; }
    double getDrudeCouplingTime() const { return drudeCouplingTime; }
    void setDrudeCouplingTime(double tau) { drudeCouplingTime = tau; }
    double getMaxDrudeDistance() const { return maxDrudeDistance; }
    void setMaxDrudeDistance(double distance) {
        if (distance < 0) throw std::runtime_error("setMaxDrudeDistance: Distance cannot be negative");   // .cpp:97-100
        maxDrudeDistance = distance;
    }
    double getStepSize() const { return stepSize; }
    void setStepSize(double dt) { stepSize = dt; }
    double getConstraintTolerance() const { return constraintTolerance; }
    void setConstraintTolerance(double tol) { constraintTolerance = tol; }
    int getDrudeStepsPerRealStep() const { return drudeStepsPerRealStep; }
    void setDrudeStepsPerRealStep(int n) { drudeStepsPerRealStep = n; }
    int getNumNHChains() const { return numNHChains; }
    void setNumNHChains(int n) { numNHChains = n; }
    int getUseDrudeNHChains() const { return useDrudeNHChains; }
    void setUseDrudeNHChains(int use) { useDrudeNHChains = use != 0; }
    bool getUseCOMTempGroup() const { return useCOMTempGroup; }
    void setUseCOMTempGroup(int use) { useCOMTempGroup = use != 0; }

    int getNumTempGroups() const { return (int)tempGroups.size(); }
    int addTempGroup() { tempGroups.push_back((int)tempGroups.size()); return (int)tempGroups.size() - 1; }   // .cpp:61-64
    int addParticleTempGroup(int tempGroup) {                                                                // .cpp:66-70
        validIndex(tempGroup, tempGroups.size());
        particleTempGroup.push_back(tempGroup);
        return (int)particleTempGroup.size() - 1;
    }
    void setParticleTempGroup(int particle, int tempGroup) {
        validIndex(particle, particleTempGroup.size());
        validIndex(tempGroup, tempGroups.size());
        particleTempGroup[particle] = tempGroup;
    }
    void getParticleTempGroup(int particle, int& tempGroup) const {
        validIndex(particle, particleTempGroup.size());
        tempGroup = particleTempGroup[particle];
    }
    int getNumResidues() const { return (int)residueInvMasses.size(); }
    double getResInvMass(int resid) const { validIndex(resid, residueInvMasses.size()); return residueInvMasses[resid]; }
    int getParticleResId(int particle) const { validIndex(particle, particleResId.size()); return particleResId[particle]; }

    /** What Context construction does (.cpp:103-160): default groups, molecule table, kernel creation.
     *  device -1 = host-only handle.  mode/precision/flags are the C ABI's. */
    void initialize(const SystemDesc& system, int device, int mode = TGNH_MODE_TGNH, int precision = TGNH_PREC_MIXED,
                    int flags = 0, double kB = 8.31446261815324e-3) {
        const int n = (int)system.mass.size();
        if (system.drudePairs.empty() && mode == TGNH_MODE_DUALNH)
            throw std::runtime_error("The System does not contain a DrudeForce");                            // .cpp:123-124
        if (particleTempGroup.empty()) {                                                                     // .cpp:127-132
            if (tempGroups.empty()) tempGroups.push_back(0);
            particleTempGroup.assign(n, 0);
        }
        else if ((int)particleTempGroup.size() != n)                                                         // .cpp:133-134
            throw std::runtime_error("Number of particles assigned with temperature groups does not match the number of system particles");
        particleResId.assign(n, -1);                                                                         // .cpp:136-141
        for (size_t i = 0; i < system.molecules.size(); i++)
            for (int p : system.molecules[i]) particleResId[p] = (int)i;
        std::vector<double> residueMasses(system.molecules.size(), 0.0);                                     // .cpp:147-153
        for (int i = 0; i < n; i++) residueMasses[particleResId[i]] += system.mass[i];
        residueInvMasses.clear();                       // (the reference appends without clearing)
        for (double m : residueMasses) residueInvMasses.push_back(1.0 / m);

        std::vector<int> pd, pp, ci, cj;
        for (auto& p : system.drudePairs) { pd.push_back(p.first); pp.push_back(p.second); }
        for (auto& c : system.constraints) { ci.push_back(c.first); cj.push_back(c.second); }
        tgnh_desc d = {};
        d.struct_size = sizeof(tgnh_desc);
        d.mode = mode; d.precision = precision; d.flags = flags; d.device = device;
        d.num_particles = n; d.padded_num_particles = (n + 31) / 32 * 32;
        d.num_pairs = (int)pd.size(); d.num_groups = getNumTempGroups(); d.num_residues = getNumResidues();
        d.num_constraints = (int)ci.size(); d.has_cm_motion_remover = system.hasCMMotionRemover;
        d.mass = system.mass.data(); d.pair_drude = pd.data(); d.pair_parent = pp.data();
        d.group = particleTempGroup.data(); d.resid = particleResId.data();
        d.constraint_i = ci.empty() ? nullptr : ci.data(); d.constraint_j = cj.empty() ? nullptr : cj.data();
        d.kB = kB; d.temperature = temperature; d.coupling_time = couplingTime;
        d.drude_temperature = drudeTemperature; d.drude_coupling_time = drudeCouplingTime; d.step_size = stepSize;
        d.drude_steps_per_real_step = drudeStepsPerRealStep; d.num_nh_chains = numNHChains;
        d.use_drude_nh_chains = useDrudeNHChains; d.use_com_temp_group = useCOMTempGroup;
        d.max_drude_distance = maxDrudeDistance;
        cleanup();
        check(tgnh_create(&d, &handle));
        paddedNumParticles = d.padded_num_particles;
    }
    void cleanup() { if (handle) { tgnh_destroy(handle); handle = nullptr; } }                               // .cpp:162-164
    tgnh_handle getHandle() const { return handle; }
    int getPaddedNumParticles() const { return paddedNumParticles; }

    void bindBuffers(void* posq, void* posqCorrection, void* velm, const void* force, void* posDelta) {
        bound(); check(tgnh_bind_buffers(handle, posq, posqCorrection, velm, force, posDelta));
    }
    /** One step with the caller's force call-out between the halves (.cpp:182-194 + kernel execute). */
    template <class ForceFn>
    void step(int steps, void* stream, ForceFn&& computeForces) {
        bound();
        for (int i = 0; i < steps; ++i) {
            check(tgnh_set_step_size(handle, stepSize));            // re-read every step, CudaDrudeTGNHKernels.cpp:292
            check(tgnh_set_drude_steps_per_real_step(handle, drudeStepsPerRealStep));
            check(tgnh_set_max_drude_distance(handle, maxDrudeDistance));
            check(tgnh_step_begin(handle, stream));
            computeForces();
            check(tgnh_step_end(handle, stream));
            isKESumValid = true;                                    // .cpp:192
        }
    }
    /** The platform's device arrays and its call-outs, as IntegrateDrudeTGNHStepKernel::execute sees them. */
    struct Buffers { void *posq, *posqCorrection, *velm; const void* force; void* posDelta; };
    struct CallOuts {
        std::function<void()> applyConstraints;          // integration.applyConstraints(tol)           Cu :363
        std::function<void()> computeVirtualSites;       // integration.computeVirtualSites()           Cu :377
        std::function<void()> calcForcesAndEnergy;       // context.calcForcesAndEnergy(true, false)    Cu :380
        std::function<void()> applyVelocityConstraints;  // integration.applyVelocityConstraints(tol)   Cu :391
    };
    /** One time step exactly as the OpenMM-HIP glue issues it (openmm_glue/HipDrudeTGNHKernels.cpp::execute, which
     *  replaces CudaDrudeTGNHKernels.cpp:284-408): the scalars the reference re-reads every step, the buffers bound
     *  again, then the fused sequence or -- with constraints -- the split one around the four call-outs. */
    void execute(void* stream, const Buffers& b, const CallOuts& co, bool hasConstraints) {
        bound();
        check(tgnh_set_step_size(handle, stepSize));                                  // Cu :292
        check(tgnh_set_drude_steps_per_real_step(handle, drudeStepsPerRealStep));     // Cu :437
        check(tgnh_set_max_drude_distance(handle, maxDrudeDistance));                 // Cu :298
        check(tgnh_bind_buffers(handle, b.posq, b.posqCorrection, b.velm, b.force, b.posDelta));
        if (!hasConstraints) {
            check(tgnh_step_begin(handle, stream));
            if (co.computeVirtualSites) co.computeVirtualSites();
            co.calcForcesAndEnergy();
            check(tgnh_step_end(handle, stream));
        }
        else {
            check(tgnh_step_begin_kick(handle, stream));                              // Cu :336-360
            co.applyConstraints();
            check(tgnh_step_begin_move(handle, stream));                              // Cu :366-376
            if (co.computeVirtualSites) co.computeVirtualSites();
            co.calcForcesAndEnergy();
            check(tgnh_step_end_kick(handle, stream));                                // Cu :384-388
            co.applyVelocityConstraints();
            check(tgnh_step_end_thermo(handle, stream));                              // Cu :394-406
        }
        isKESumValid = true;                                                          // .cpp:192
    }
    /** The device's status word; throws for what the library treats as a failure (see tgnh_get_status_flags). */
    uint32_t checkStatus(void* stream) {
        bound();
        uint32_t flags = 0;
        check(tgnh_get_status_flags(handle, stream, &flags));
        return flags;
    }
    void stateChanged() { isKESumValid = false; if (handle) check(tgnh_state_changed(handle)); }             // .cpp:166-170
    double computeKineticEnergy(void* stream) {                                                              // .cpp:178-180
        bound();
        double ke = 0;
        check(tgnh_get_kinetic_energy(handle, isKESumValid, stream, &ke));
        return ke;
    }

private:
    static void validIndex(long i, size_t n) { if (i < 0 || (size_t)i >= n) throw std::runtime_error("Index out of range"); }
    static void check(tgnh_status rc) { if (rc != TGNH_OK) throw std::runtime_error(tgnh_last_error()); }
    void bound() const { if (!handle) throw std::runtime_error("This Integrator is not bound to a context!"); }   // .cpp:183-184
    double temperature, couplingTime, drudeTemperature, drudeCouplingTime, maxDrudeDistance, stepSize, constraintTolerance;
    int drudeStepsPerRealStep, numNHChains;
    bool useDrudeNHChains, useCOMTempGroup, isKESumValid = false;
    std::vector<int> particleTempGroup, tempGroups, particleResId;
    std::vector<double> residueInvMasses;
    tgnh_handle handle = nullptr;
    int paddedNumParticles = 0;
};

}  // namespace drudetgnh
#endif
