// XML proxy of DrudeTGNHIntegrator (see the header).  Written against OpenMM's serialization API (SerializationNode's
// set/get{Int,Double,Bool}Property, createChildNode, getChildren) and the reference's unmodified API class.
#include "openmm/serialization/DrudeTGNHIntegratorProxy.h"
#include "openmm/serialization/SerializationNode.h"
#include "openmm/DrudeTGNHIntegrator.h"
#include "openmm/OpenMMException.h"
#include <string>

using namespace OpenMM;
using namespace std;

DrudeTGNHThermostatStore::Reader DrudeTGNHThermostatStore::reader = nullptr;
map<const DrudeTGNHIntegrator*, DrudeTGNHThermostatState> DrudeTGNHThermostatStore::parked;

void DrudeTGNHThermostatStore::setReader(Reader r) { reader = r; }
bool DrudeTGNHThermostatStore::read(const DrudeTGNHIntegrator* integrator, DrudeTGNHThermostatState& state) {
    return reader != nullptr && reader(integrator, state);
}
void DrudeTGNHThermostatStore::put(const DrudeTGNHIntegrator* integrator, const DrudeTGNHThermostatState& state) {
    parked[integrator] = state;
}
bool DrudeTGNHThermostatStore::take(const DrudeTGNHIntegrator* integrator, DrudeTGNHThermostatState& state) {
    map<const DrudeTGNHIntegrator*, DrudeTGNHThermostatState>::iterator it = parked.find(integrator);
    if (it == parked.end())
        return false;
    state = it->second;
    parked.erase(it);
    return true;
}

DrudeTGNHIntegratorProxy::DrudeTGNHIntegratorProxy() : SerializationProxy("DrudeTGNHIntegrator") {
}

static void writeArray(SerializationNode& parent, const string& name, const vector<double>& values) {
    SerializationNode& node = parent.createChildNode(name);
    for (size_t i = 0; i < values.size(); i++)
        node.createChildNode("Value").setDoubleProperty("v", values[i]);
}

static vector<double> readArray(const SerializationNode& node) {
    vector<double> values;
    for (size_t i = 0; i < node.getChildren().size(); i++)
        values.push_back(node.getChildren()[i].getDoubleProperty("v"));
    return values;
}

// The reference's nine properties (serialization/src/DrudeTGNHIntegratorProxy.cpp:43-55): the same names and the same version, so
// that files of either proxy are read by the other.
namespace {
struct RealProperty { const char* name; double (DrudeTGNHIntegrator::*get)() const; };
struct CountProperty { const char* name; int (DrudeTGNHIntegrator::*get)() const; };
const RealProperty realProperties[] = {
    {"stepSize", &DrudeTGNHIntegrator::getStepSize},          {"constraintTolerance", &DrudeTGNHIntegrator::getConstraintTolerance},
    {"temperature", &DrudeTGNHIntegrator::getTemperature},    {"couplingTime", &DrudeTGNHIntegrator::getCouplingTime},
    {"drudeTemperature", &DrudeTGNHIntegrator::getDrudeTemperature}, {"drudeCouplingTime", &DrudeTGNHIntegrator::getDrudeCouplingTime}};
const CountProperty countProperties[] = {
    {"drudeStepsPerRealStep", &DrudeTGNHIntegrator::getDrudeStepsPerRealStep}, {"numNHChains", &DrudeTGNHIntegrator::getNumNHChains}};
const int proxyVersion = 1;
}

void DrudeTGNHIntegratorProxy::serialize(const void* object, SerializationNode& node) const {
    const DrudeTGNHIntegrator& integrator = *static_cast<const DrudeTGNHIntegrator*>(object);
    node.setIntProperty("version", proxyVersion);
    for (const RealProperty& p : realProperties)
        node.setDoubleProperty(p.name, (integrator.*p.get)());
    for (const CountProperty& p : countProperties)
        node.setIntProperty(p.name, (integrator.*p.get)());
    node.setIntProperty("useDrudeNHChains", integrator.getUseDrudeNHChains() ? 1 : 0);
    // what that proxy drops (a version-1 reader of the reference ignores properties and children it does not ask for)
    node.setDoubleProperty("maxDrudeDistance", integrator.getMaxDrudeDistance());
    node.setIntProperty("useCOMTempGroup", integrator.getUseCOMTempGroup());
    if (integrator.getNumTempGroups() > 0) {
        SerializationNode& groups = node.createChildNode("TempGroups");
        groups.setIntProperty("count", integrator.getNumTempGroups());
        // The API class has no getNumParticleTempGroups(): the table is walked until its range check throws
        // (ASSERT_VALID_INDEX, openmmapi/src/DrudeTGNHIntegrator.cpp:78-81)
        for (int particle = 0; ; particle++) {
            int group;
            try {
                integrator.getParticleTempGroup(particle, group);
            }
            catch (const OpenMMException&) {
                break;
            }
            groups.createChildNode("Particle").setIntProperty("group", group);
        }
    }
    DrudeTGNHThermostatState state;
    if (DrudeTGNHThermostatStore::read(&integrator, state)) {
        SerializationNode& thermo = node.createChildNode("ThermostatState");
        thermo.setDoubleProperty("time", state.time);
        thermo.setStringProperty("stepCount", to_string(state.stepCount));       // (64 bits: not through an int property)
        writeArray(thermo, "eta", state.eta);
        writeArray(thermo, "etaDot", state.etaDot);
        writeArray(thermo, "etaDotDot", state.etaDotDot);
    }
}

void* DrudeTGNHIntegratorProxy::deserialize(const SerializationNode& node) const {
    const int version = node.getIntProperty("version");
    if (version != proxyVersion)
        throw OpenMMException("Unsupported version number");                     // (the reference's message, :58-59)
    const double temperature = node.getDoubleProperty("temperature"), couplingTime = node.getDoubleProperty("couplingTime");
    const double drudeTemperature = node.getDoubleProperty("drudeTemperature"), drudeCouplingTime = node.getDoubleProperty("drudeCouplingTime");
    const int substeps = node.getIntProperty("drudeStepsPerRealStep"), links = node.getIntProperty("numNHChains");
    const bool drudeChains = node.getBoolProperty("useDrudeNHChains");
    DrudeTGNHIntegrator* integrator = new DrudeTGNHIntegrator(temperature, couplingTime, drudeTemperature, drudeCouplingTime,
                                                              node.getDoubleProperty("stepSize"), substeps, links, drudeChains);
    try {
        integrator->setConstraintTolerance(node.getDoubleProperty("constraintTolerance"));
        // files written by the reference's proxy have none of the following: the constructor's defaults stay
        if (node.hasProperty("maxDrudeDistance"))
            integrator->setMaxDrudeDistance(node.getDoubleProperty("maxDrudeDistance"));
        if (node.hasProperty("useCOMTempGroup"))
            integrator->setUseCOMTempGroup(node.getIntProperty("useCOMTempGroup"));
        for (size_t i = 0; i < node.getChildren().size(); i++) {
            const SerializationNode& child = node.getChildren()[i];
            if (child.getName() == "TempGroups") {
                for (int g = 0; g < child.getIntProperty("count"); g++)
                    integrator->addTempGroup();
                for (size_t p = 0; p < child.getChildren().size(); p++)
                    integrator->addParticleTempGroup(child.getChildren()[p].getIntProperty("group"));
            }
            else if (child.getName() == "ThermostatState") {
                DrudeTGNHThermostatState state;
                state.time = child.getDoubleProperty("time");
                state.stepCount = stoll(child.getStringProperty("stepCount"));
                for (size_t a = 0; a < child.getChildren().size(); a++) {
                    const SerializationNode& arr = child.getChildren()[a];
                    if (arr.getName() == "eta") state.eta = readArray(arr);
                    else if (arr.getName() == "etaDot") state.etaDot = readArray(arr);
                    else if (arr.getName() == "etaDotDot") state.etaDotDot = readArray(arr);
                }
                DrudeTGNHThermostatStore::put(integrator, state);
            }
        }
    }
    catch (...) {
        delete integrator;
        throw;
    }
    return integrator;
}
