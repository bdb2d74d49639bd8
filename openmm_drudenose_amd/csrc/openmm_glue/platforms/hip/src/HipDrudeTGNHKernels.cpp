// OpenMM-HIP glue (not compiled here, see HipDrudeTGNHKernels.h).  Everything with arithmetic in it lives
// behind the C ABI; this file only translates OpenMM objects into tgnh_desc and device pointers.
#include "HipDrudeTGNHKernels.h"
#include "openmm/AndersenThermostat.h"
#include "openmm/CMMotionRemover.h"
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"
#include "SimTKOpenMMRealType.h"
#include "openmm/KernelFactory.h"
#include "openmm/hip/HipPlatform.h"
#ifdef DRUDETGNH_THERMOSTAT_CHECKPOINT
#include "openmm/serialization/DrudeTGNHIntegratorProxy.h"
#include <map>
#endif
#include <cstring>
#include <typeinfo>
#include <vector>

using namespace OpenMM;
using namespace std;

void HipIntegrateDrudeTGNHStepKernel::check(tgnh_status rc) const {
    if (rc != TGNH_OK)
        throw OpenMMException(tgnh_last_error());       // same channel as the reference's throws
}

#ifdef DRUDETGNH_THERMOSTAT_CHECKPOINT
// Which kernel integrates for which integrator object: what the serialization proxy's reader looks up (DrudeTGNHThermostatStore,
// serialization/include/openmm/serialization/DrudeTGNHIntegratorProxy.h).  One context per integrator (OpenMM's rule).
static std::map<const DrudeTGNHIntegrator*, HipIntegrateDrudeTGNHStepKernel*> kernelOf;

bool HipIntegrateDrudeTGNHStepKernel::readThermostat(DrudeTGNHThermostatState& state) {
    ContextSelector selector(cu);
    void* stream = (void*) cu.getCurrentStream();
    std::vector<double>* arrays[3] = {&state.eta, &state.etaDot, &state.etaDotDot};
    for (int which = 0; which < 3; which++) {
        int len = 0;
        check(tgnh_get_thermostat_len(handle, which, &len));
        arrays[which]->resize(len);
        check(tgnh_get_thermostat_state(handle, which, stream, arrays[which]->data()));
    }
    int64_t steps = 0;
    check(tgnh_get_time(handle, &state.time, &steps));
    state.stepCount = steps;
    return true;
}

static bool readThermostatOf(const DrudeTGNHIntegrator* integrator, DrudeTGNHThermostatState& state) {
    std::map<const DrudeTGNHIntegrator*, HipIntegrateDrudeTGNHStepKernel*>::iterator it = kernelOf.find(integrator);
    return it != kernelOf.end() && it->second->readThermostat(state);
}
#endif

HipIntegrateDrudeTGNHStepKernel::~HipIntegrateDrudeTGNHStepKernel() {
#ifdef DRUDETGNH_THERMOSTAT_CHECKPOINT
    for (std::map<const DrudeTGNHIntegrator*, HipIntegrateDrudeTGNHStepKernel*>::iterator it = kernelOf.begin(); it != kernelOf.end(); )
        if (it->second == this) kernelOf.erase(it++); else ++it;
#endif
    if (handle != nullptr) {
        ContextSelector selector(cu);
        tgnh_destroy(handle);
    }
}

void HipIntegrateDrudeTGNHStepKernel::initialize(const System& system, const DrudeTGNHIntegrator& integrator, const DrudeForce& force) {
    cu.getPlatformData().initializeContexts(system);
    ContextSelector selector(cu);
    const int numParticles = system.getNumParticles();
    vector<double> mass(numParticles);
    vector<int> group(numParticles), resid(numParticles);
    for (int i = 0; i < numParticles; i++) {
        mass[i] = system.getParticleMass(i);
        integrator.getParticleTempGroup(i, group[i]);
        resid[i] = integrator.getParticleResId(i);
    }
    vector<int> pairDrude(force.getNumParticles()), pairParent(force.getNumParticles());
    for (int i = 0; i < force.getNumParticles(); i++) {
        int p, p1, p2, p3, p4;
        double charge, polarizability, aniso12, aniso34;
        force.getParticleParameters(i, p, p1, p2, p3, p4, charge, polarizability, aniso12, aniso34);
        pairDrude[i] = p;
        pairParent[i] = p1;
    }
    numConstraints = system.getNumConstraints();
    vector<int> ci(numConstraints), cj(numConstraints);
    for (int i = 0; i < numConstraints; i++) {
        double distance;
        system.getConstraintParameters(i, ci[i], cj[i], distance);
    }
    bool hasCMM = false, onlyKnownForces = true;
    // DRUDETGNH_TRUST_STATE_CHANGED is taken only for a System whose every Force is of a type KNOWN not to write velocities in
    // updateContextState (an allow-list by class name, as the reference sniffs its barostat, DrudeTGNHIntegrator.cpp:117-121):
    // CMMotionRemover and AndersenThermostat do so silently (no stateChanged), and a plugin Force this file has never heard of
    // might -- with a deny-list such a Force would leave the begin half on a stale sum with no error.
    static const char* const velocityNeutral[] = {"HarmonicBondForce", "HarmonicAngleForce", "PeriodicTorsionForce", "RBTorsionForce",
        "CMAPTorsionForce", "NonbondedForce", "CustomNonbondedForce", "CustomBondForce", "CustomAngleForce", "CustomTorsionForce",
        "CustomCompoundBondForce", "CustomCentroidBondForce", "CustomExternalForce", "CustomHbondForce", "CustomManyParticleForce",
        "CustomGBForce", "CustomCVForce", "GBSAOBCForce", "GayBerneForce", "DrudeForce", "MonteCarloBarostat",
        "MonteCarloAnisotropicBarostat", "MonteCarloMembraneBarostat", "MonteCarloFlexibleBarostat"};
    for (int i = 0; i < system.getNumForces(); i++) {
        if (dynamic_cast<const CMMotionRemover*>(&system.getForce(i)) != nullptr)
            hasCMM = true;
        const char* type = typeid(system.getForce(i)).name();          // (mangled: the class name is its tail, "...<len>Name" + "E")
        bool known = false;
        for (size_t k = 0; k < sizeof(velocityNeutral) / sizeof(velocityNeutral[0]); k++) {
            const char* at = strstr(type, velocityNeutral[k]);
            if (at != nullptr && (at[strlen(velocityNeutral[k])] == 'E' || at[strlen(velocityNeutral[k])] == '\0'))
                known = true;
        }
        if (!known)
            onlyKnownForces = false;
    }
    (void) onlyKnownForces;                                // (read under DRUDETGNH_TRUST_STATE_CHANGED only)

    tgnh_desc d = {};
    d.struct_size = sizeof(tgnh_desc);
    d.mode = TGNH_MODE_TGNH;                               // the GPU platform's semantics (temperature groups + COM)
    d.precision = cu.getUseDoublePrecision() ? TGNH_PREC_DOUBLE : (cu.getUseMixedPrecision() ? TGNH_PREC_MIXED : TGNH_PREC_SINGLE);
    d.flags = 0;                                           // OpenMM may touch velocities between steps: keep the plain pass structure
#ifdef DRUDETGNH_RESIDENT_STEP
    // ... with each thermostat half as one launch (velocities still never lag).  Needs this context to have the device to
    // itself while it steps: off by default (CMake option DRUDETGNH_RESIDENT_STEP), because several OpenMM contexts may share a GPU
    d.flags = TGNH_FLAG_RESIDENT_STEP;
#endif
#ifdef DRUDETGNH_TRUST_STATE_CHANGED
    // The begin half of a step starts its chain from the kinetic energies the last end half's chain left (s^2 KE, Cu :574)
    // instead of summing them again (Cu :474-488): one pass over the velocities less per step.  Exact as long as nothing writes
    // velocities between two steps without the integrator hearing of it: Context::setVelocities and friends reach
    // DrudeTGNHIntegrator::stateChanged (DrudeTGNHIntegrator.cpp:166-170), which execute() below forwards through
    // integrator.isKineticEnergySumValid() -- the ONE-LINE accessor this option needs in DrudeTGNHIntegrator.h (INTEGRATION.md
    // section 3); Forces are vetted by the allow-list above.  Context::applyVelocityConstraints (a user call between steps) writes
    // velocities WITHOUT stateChanged: a caller that uses it with this option must follow it with setVelocities(getVelocities())
    // or not build with this option.
    trustStateChanged = onlyKnownForces;
    if (trustStateChanged)
        d.flags |= TGNH_FLAG_TRUST_STATE_CHANGED;
#endif
    d.device = cu.getDeviceIndex();
    d.num_particles = numParticles;
    d.padded_num_particles = cu.getPaddedNumAtoms();
    d.num_pairs = force.getNumParticles();
    d.num_groups = integrator.getNumTempGroups();
    d.num_residues = integrator.getNumResidues();
    d.num_constraints = numConstraints;
    d.has_cm_motion_remover = hasCMM;
    d.mass = mass.data();
    d.pair_drude = pairDrude.data();
    d.pair_parent = pairParent.data();
    d.group = group.data();
    d.resid = resid.data();
    d.constraint_i = ci.data();
    d.constraint_j = cj.data();
    d.kB = BOLTZ;
    d.temperature = integrator.getTemperature();
    d.coupling_time = integrator.getCouplingTime();
    d.drude_temperature = integrator.getDrudeTemperature();
    d.drude_coupling_time = integrator.getDrudeCouplingTime();
    d.step_size = integrator.getStepSize();
    d.drude_steps_per_real_step = integrator.getDrudeStepsPerRealStep();
    d.num_nh_chains = integrator.getNumNHChains();
    d.use_drude_nh_chains = integrator.getUseDrudeNHChains();
    d.use_com_temp_group = integrator.getUseCOMTempGroup();
    d.max_drude_distance = integrator.getMaxDrudeDistance();
    check(tgnh_create(&d, &handle));
#ifdef DRUDETGNH_THERMOSTAT_CHECKPOINT
    // an integrator that came out of the XML proxy with a ThermostatState child continues where the checkpoint was taken
    // (thermostat arrays + clock: the trajectory's bits are a function of state and step counter, include/drude_tgnh.h)
    kernelOf[&integrator] = this;
    DrudeTGNHThermostatStore::setReader(readThermostatOf);
    DrudeTGNHThermostatState state;
    if (DrudeTGNHThermostatStore::take(&integrator, state)) {
        void* stream = (void*) cu.getCurrentStream();
        const std::vector<double>* arrays[3] = {&state.eta, &state.etaDot, &state.etaDotDot};
        for (int which = 0; which < 3; which++) {
            int len = 0;
            check(tgnh_get_thermostat_len(handle, which, &len));
            if ((int) arrays[which]->size() != len)
                throw OpenMMException("DrudeTGNHIntegrator: the checkpointed thermostat state does not fit this System (numNHChains / temperature groups changed?)");
            check(tgnh_set_thermostat_state(handle, which, stream, arrays[which]->data()));
        }
        check(tgnh_set_time(handle, state.time, state.stepCount));
    }
#endif
}

void HipIntegrateDrudeTGNHStepKernel::execute(ContextImpl& context, const DrudeTGNHIntegrator& integrator) {
    ContextSelector selector(cu);
    IntegrationUtilities& integration = cu.getIntegrationUtilities();
    void* stream = (void*) cu.getCurrentStream();
    // values the reference re-reads every step (CudaDrudeTGNHKernels.cpp:292, :298, :437)
    check(tgnh_set_step_size(handle, integrator.getStepSize()));
    check(tgnh_set_drude_steps_per_real_step(handle, integrator.getDrudeStepsPerRealStep()));
    check(tgnh_set_max_drude_distance(handle, integrator.getMaxDrudeDistance()));
    // OpenMM may reallocate nothing here, but reordering keeps the same arrays: bind every step (cheap, no launch)
    check(tgnh_bind_buffers(handle, (void*) cu.getPosq().getDevicePointer(),
                            cu.getUseMixedPrecision() ? (void*) cu.getPosqCorrection().getDevicePointer() : nullptr,
                            (void*) cu.getVelm().getDevicePointer(), (const void*) cu.getForce().getDevicePointer(),
                            (void*) integration.getPosDelta().getDevicePointer()));
#ifdef DRUDETGNH_TRUST_STATE_CHANGED
    if (trustStateChanged && !integrator.isKineticEnergySumValid())     // stateChanged since the last step (DrudeTGNHIntegrator.cpp:166-170, :192)
        check(tgnh_state_changed(handle));
#endif
    if (cu.getAtomsWereReordered())                         // CudaDrudeTGNHKernels.cpp:344-347
        context.calcForcesAndEnergy(true, false);
    if (numConstraints == 0) {
        check(tgnh_step_begin(handle, stream));             // thermostat half, rescale, kick, drift, hard wall
        integration.computeVirtualSites();                  // :377
        context.calcForcesAndEnergy(true, false);           // :380
        check(tgnh_step_end(handle, stream));               // kick, thermostat half, rescale
    }
    else {
        check(tgnh_step_begin_kick(handle, stream));        // :336-360
        integration.applyConstraints(integrator.getConstraintTolerance());          // :363
        check(tgnh_step_begin_move(handle, stream));        // :366-376
        integration.computeVirtualSites();
        context.calcForcesAndEnergy(true, false);
        check(tgnh_step_end_kick(handle, stream));          // :384-388
        integration.applyVelocityConstraints(integrator.getConstraintTolerance());  // :391
        check(tgnh_step_end_thermo(handle, stream));        // :394-402
    }
    cu.setTime(cu.getTime()+integrator.getStepSize());      // :405-407
    cu.setStepCount(cu.getStepCount()+1);
    cu.reorderAtoms();
}

double HipIntegrateDrudeTGNHStepKernel::computeKineticEnergy(ContextImpl& context, const DrudeTGNHIntegrator& integrator, bool isKESumValid) {
    ContextSelector selector(cu);
    if (!isKESumValid)                                      // CudaDrudeTGNHKernels.cpp:655-656
        return cu.getIntegrationUtilities().computeKineticEnergy(0);
    double ke;
    check(tgnh_get_kinetic_energy(handle, 1, (void*) cu.getCurrentStream(), &ke));
    return ke;
}

// ---------------------------------------------------------------------------------------------------------------
// Plugin entry points for the OpenMM "HIP" platform.  OpenMM dlopen()s every library in lib/plugins and looks these
// C symbols up; they are the ones the reference's CUDA plugin exports for its platform
// (platforms/cuda/src/CudaDrudeTGNHKernelFactory.cpp:37-59).
// ---------------------------------------------------------------------------------------------------------------
namespace {
class HipDrudeTGNHKernelFactory : public KernelFactory {
public:
    KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const {
        HipContext& cu = *static_cast<HipPlatform::PlatformData*>(context.getPlatformData())->contexts[0];
        if (name == IntegrateDrudeTGNHStepKernel::Name())
            return new HipIntegrateDrudeTGNHStepKernel(name, platform, cu);
        throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '")+name+"'").c_str());
    }
};
}

extern "C" OPENMM_EXPORT void registerPlatforms() {
}

extern "C" OPENMM_EXPORT void registerKernelFactories() {
    try {
        Platform& platform = Platform::getPlatformByName("HIP");
        platform.registerKernelFactory(IntegrateDrudeTGNHStepKernel::Name(), new HipDrudeTGNHKernelFactory());
    }
    catch (std::exception& ex) {
        // no HIP platform in this OpenMM: nothing to register (as the reference does for CUDA)
    }
}

extern "C" OPENMM_EXPORT void registerDrudeTGNHHipKernelFactories() {
    try {
        Platform::getPlatformByName("HIP");
    }
    catch (...) {
        Platform::registerPlatform(new HipPlatform());
    }
    registerKernelFactories();
}
