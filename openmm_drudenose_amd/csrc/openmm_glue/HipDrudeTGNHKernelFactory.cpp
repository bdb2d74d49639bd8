// Plugin entry points for the OpenMM "HIP" platform (not compiled here, see HipDrudeTGNHKernels.h).
// Same three C symbols the reference's CUDA plugin exports (platforms/cuda/src/CudaDrudeTGNHKernelFactory.cpp:37-59),
// looked up by OpenMM after dlopen() of every library in lib/plugins.
#include "HipDrudeTGNHKernels.h"
#include "openmm/KernelFactory.h"
#include "openmm/OpenMMException.h"
#include "openmm/hip/HipPlatform.h"
#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

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
