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

void DrudeTGNHIntegratorProxy::serialize(const void* object, SerializationNode& node) const {
    node.setIntProperty("version", 1);
    const DrudeTGNHIntegrator& integrator = *reinterpret_cast<const DrudeTGNHIntegrator*>(object);
    // the reference's nine (serialization/src/DrudeTGNHIntegratorProxy.cpp:43-55): same names, same version -- files of either
    // proxy are read by the other
    node.setDoubleProperty("stepSize", integrator.getStepSize());
    node.setDoubleProperty("constraintTolerance", integrator.getConstraintTolerance());
    node.setDoubleProperty("temperature", integrator.getTemperature());
    node.setDoubleProperty("couplingTime", integrator.getCouplingTime());
    node.setDoubleProperty("drudeTemperature", integrator.getDrudeTemperature());
    node.setDoubleProperty("drudeCouplingTime", integrator.getDrudeCouplingTime());
    node.setIntProperty("drudeStepsPerRealStep", integrator.getDrudeStepsPerRealStep());
    node.setIntProperty("numNHChains", integrator.getNumNHChains());
    node.setIntProperty("useDrudeNHChains", integrator.getUseDrudeNHChains());
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
    if (node.getIntProperty("version") != 1)
        throw OpenMMException("Unsupported version number");
    DrudeTGNHIntegrator *integrator = new DrudeTGNHIntegrator(node.getDoubleProperty("temperature"),
            node.getDoubleProperty("couplingTime"), node.getDoubleProperty("drudeTemperature"),
            node.getDoubleProperty("drudeCouplingTime"), node.getDoubleProperty("stepSize"),
            node.getIntProperty("drudeStepsPerRealStep"), node.getIntProperty("numNHChains"),
            node.getBoolProperty("useDrudeNHChains"));
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
