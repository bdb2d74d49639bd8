#ifndef HIP_DRUDE_TGNH_KERNELS_H_
#define HIP_DRUDE_TGNH_KERNELS_H_
// OpenMM-HIP glue: IntegrateDrudeTGNHStepKernel implemented on the C ABI of include/drude_tgnh.h.
// NOT compiled in this repository (OpenMM is absent from the image); INTEGRATION.md says where it goes.
// It replaces platforms/cuda/src/CudaDrudeTGNHKernels.{h,cpp} of scychon/openmm_drudeNose and is written
// against OpenMM >= 8.2 with the HIP platform (openmm/hip/HipContext.h).
#include "openmm/DrudeTGNHKernels.h"      // the reference's own abstract kernel (openmmapi/include/openmm/DrudeTGNHKernels.h:48-74)
#include "openmm/hip/HipContext.h"
#include "drude_tgnh.h"

namespace OpenMM {

class HipIntegrateDrudeTGNHStepKernel : public IntegrateDrudeTGNHStepKernel {
public:
    HipIntegrateDrudeTGNHStepKernel(std::string name, const Platform& platform, HipContext& cu)
        : IntegrateDrudeTGNHStepKernel(name, platform), cu(cu), handle(nullptr) {}
    ~HipIntegrateDrudeTGNHStepKernel();
    void initialize(const System& system, const DrudeTGNHIntegrator& integrator, const DrudeForce& force);
    void execute(ContextImpl& context, const DrudeTGNHIntegrator& integrator);
    double computeKineticEnergy(ContextImpl& context, const DrudeTGNHIntegrator& integrator, bool isKESumValid);
#ifdef DRUDETGNH_THERMOSTAT_CHECKPOINT
    bool readThermostat(struct DrudeTGNHThermostatState& state);     // for the serialization proxy (DrudeTGNHThermostatStore)
#endif
private:
    void check(tgnh_status rc) const;
    HipContext& cu;
    tgnh_handle handle;
    int numConstraints;
    bool trustStateChanged = false;
};

}  // namespace OpenMM
#endif
