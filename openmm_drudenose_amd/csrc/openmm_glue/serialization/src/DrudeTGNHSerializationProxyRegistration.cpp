// Registers the proxy when the library is loaded: the counterpart of the reference's
// serialization/src/DrudeTGNHSerializationProxyRegistration.cpp:52-65 (same entry point name, so a build that links this file
// instead of the reference's has the same symbol; Linux only -- the MI355X build has no Windows target).
#include <dlfcn.h>

#include "openmm/OpenMMException.h"
#include "openmm/DrudeTGNHIntegrator.h"
#include "openmm/serialization/SerializationProxy.h"
#include "openmm/serialization/DrudeTGNHIntegratorProxy.h"
#include <typeinfo>

extern "C" void __attribute__((constructor)) registerDrudeTGNHSerializationProxies();

using namespace OpenMM;

extern "C" OPENMM_EXPORT void registerDrudeTGNHSerializationProxies() {
    SerializationProxy::registerProxy(typeid(DrudeTGNHIntegrator), new DrudeTGNHIntegratorProxy());
}
