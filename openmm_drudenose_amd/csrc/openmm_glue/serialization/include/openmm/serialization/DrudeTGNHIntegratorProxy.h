#ifndef OPENMM_DRUDE_TGNH_INTEGRATOR_PROXY_HIP_H_
#define OPENMM_DRUDE_TGNH_INTEGRATOR_PROXY_HIP_H_
// XML proxy of DrudeTGNHIntegrator for the plugin tree of the MI355X build.  Takes the place of the reference's
// serialization/include/openmm/serialization/DrudeTGNHIntegratorProxy.h:45-50 (same class name, same registration entry point,
// so it is a drop-in for libOpenMMDrudeTGNH's serialization part): the nine properties of
// serialization/src/DrudeTGNHIntegratorProxy.cpp:43-55 under the same names and version, plus what that proxy drops --
// maxDrudeDistance, useCOMTempGroup, the temperature-group table -- and, when a context's kernel has registered a reader
// (DrudeTGNHThermostatStore), the thermostat arrays and the clock, which the reference cannot checkpoint at all.
// NOT compiled in this repository (OpenMM is absent from the image): tests/test_glue_syntax.py runs it through g++ -fsyntax-only.
#include "openmm/serialization/SerializationProxy.h"
#include <map>
#include <vector>

namespace OpenMM {

class DrudeTGNHIntegrator;

class OPENMM_EXPORT DrudeTGNHIntegratorProxy : public SerializationProxy {
public:
    DrudeTGNHIntegratorProxy();
    void serialize(const void* object, SerializationNode& node) const;
    void* deserialize(const SerializationNode& node) const;
};

// Thermostat variables and clock of the kernel that integrates for an integrator object: eta, etaDot, etaDotDot in the layout of
// tgnh_get_thermostat_state (include/drude_tgnh.h), time and step count as tgnh_get_time returns them.
struct DrudeTGNHThermostatState {
    std::vector<double> eta, etaDot, etaDotDot;
    double time = 0;
    long long stepCount = 0;
};

// Process-wide hand-over between the proxy and the platform kernels, keyed by integrator object.  The integrator's API class
// (openmmapi/include/openmm/DrudeTGNHIntegrator.h) holds no thermostat state -- its kernels do -- so:
//   serialize:   the proxy asks the reader a platform plugin registered (the HIP kernel: tgnh_get_thermostat_state of the handle
//                it created for that integrator); no reader, or no context yet: the node simply has no ThermostatState child;
//   deserialize: the proxy parks the state it read; the kernel's initialize() takes it (tgnh_set_thermostat_state, tgnh_set_time).
class OPENMM_EXPORT DrudeTGNHThermostatStore {
public:
    typedef bool (*Reader)(const DrudeTGNHIntegrator* integrator, DrudeTGNHThermostatState& state);
    static void setReader(Reader reader);
    static bool read(const DrudeTGNHIntegrator* integrator, DrudeTGNHThermostatState& state);
    static void put(const DrudeTGNHIntegrator* integrator, const DrudeTGNHThermostatState& state);
    static bool take(const DrudeTGNHIntegrator* integrator, DrudeTGNHThermostatState& state);
private:
    static Reader reader;
    static std::map<const DrudeTGNHIntegrator*, DrudeTGNHThermostatState> parked;
};

} // namespace OpenMM

#endif /*OPENMM_DRUDE_TGNH_INTEGRATOR_PROXY_HIP_H_*/
