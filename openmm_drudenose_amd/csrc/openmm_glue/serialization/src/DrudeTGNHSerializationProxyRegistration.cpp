// Registers the XML proxy of DrudeTGNHIntegrator when the library is loaded, and again -- harmlessly -- when somebody calls the entry
// point by name: the counterpart of the reference's serialization/src/DrudeTGNHSerializationProxyRegistration.cpp:52-65 (same entry
// point, so a build that links this file instead of the reference's exports the same symbol; Linux only -- the MI355X build has no
// Windows target, hence no DllMain).
#include <typeinfo>

#include "openmm/DrudeTGNHIntegrator.h"
#include "openmm/OpenMMException.h"
#include "openmm/serialization/DrudeTGNHIntegratorProxy.h"
#include "openmm/serialization/SerializationProxy.h"

extern "C" OPENMM_EXPORT void registerDrudeTGNHSerializationProxies() {
    static bool registered = false;                 // (a second registration would leak the first proxy)
    if (registered)
        return;
    registered = true;
    OpenMM::SerializationProxy::registerProxy(typeid(OpenMM::DrudeTGNHIntegrator), new OpenMM::DrudeTGNHIntegratorProxy());
}

namespace {
struct RegisterAtLoad {
    RegisterAtLoad() { registerDrudeTGNHSerializationProxies(); }
} registerAtLoad;
}
