// NOT OpenMM: bodies for shim_common.h's declarations (compiled with -DTGNH_SHIM_FUNCTIONAL), a miniature runtime that behaves
// towards the glue as the OpenMM classes of the same names do -- as far as the glue touches them: containers with getters, a
// platform registry, call-outs forwarded to what the test installs.  tests/cpp/test_glue_linked.cpp links the REAL glue sources
// (openmm_drudenose_amd/csrc/openmm_glue/platforms/hip/src/HipDrudeTGNHKernels.cpp, .../serialization/src/*.cpp) against it.
// What each function does in OpenMM is cited where it is not obvious; nothing here is taken from OpenMM's or the reference's sources.
#ifndef TGNH_SHIM_FUNCTIONAL
#error "compile with -DTGNH_SHIM_FUNCTIONAL"
#endif
#include "shim_common.h"
#include <cstdio>
#include <cstdlib>

namespace OpenMM {

OpenMMException::OpenMMException(const std::string& message) : shimMessage(message) {}
const char* OpenMMException::what() const noexcept { return shimMessage.c_str(); }

Force::~Force() {}

int System::getNumParticles() const { return (int)shimMass.size(); }
double System::getParticleMass(int index) const { return shimMass.at(index); }
int System::getNumConstraints() const { return (int)shimConstraints.size(); }
void System::getConstraintParameters(int index, int& particle1, int& particle2, double& distance) const {
    const ShimConstraint& c = shimConstraints.at(index);
    particle1 = c.a; particle2 = c.b; distance = c.d;
}
int System::getNumForces() const { return (int)shimForces.size(); }
const Force& System::getForce(int index) const { return *shimForces.at(index); }

int DrudeForce::getNumParticles() const { return (int)shimDrude.size(); }
void DrudeForce::getParticleParameters(int index, int& particle, int& particle1, int& particle2, int& particle3, int& particle4,
                                       double& charge, double& polarizability, double& aniso12, double& aniso34) const {
    particle = shimDrude.at(index); particle1 = shimParent.at(index);
    particle2 = particle3 = particle4 = -1;                 // isotropic
    charge = -1.0; polarizability = 1e-3; aniso12 = aniso34 = 1.0;
}

// the API class (the reference's openmmapi/src/DrudeTGNHIntegrator.cpp: setters, getters, the temperature-group table with its range
// check :78-81, the non-negative hard-wall distance :98-99)
DrudeTGNHIntegrator::DrudeTGNHIntegrator(double temperature, double couplingTime, double drudeTemperature, double drudeCouplingTime,
                                         double stepSize, int drudeStepsPerRealStep, int numNHChains, bool useDrudeNHChains, bool useCOMTempGroup)
    : shimT(temperature), shimTau(couplingTime), shimDrudeT(drudeTemperature), shimDrudeTau(drudeCouplingTime), shimDt(stepSize),
      shimDrudeSteps(drudeStepsPerRealStep), shimChains(numNHChains), shimDrudeChains(useDrudeNHChains), shimCOM(useCOMTempGroup) {}
void DrudeTGNHIntegrator::setConstraintTolerance(double tol) { shimTol = tol; }
void DrudeTGNHIntegrator::setMaxDrudeDistance(double distance) {
    if (distance < 0) throw OpenMMException("setMaxDrudeDistance: Distance cannot be negative");
    shimMaxDist = distance;
}
void DrudeTGNHIntegrator::setUseCOMTempGroup(int useCOMGroup) { shimCOM = useCOMGroup != 0; }
int DrudeTGNHIntegrator::addTempGroup() { return shimNumGroups++; }
int DrudeTGNHIntegrator::addParticleTempGroup(int tempGroup) {
    if (tempGroup < 0 || tempGroup >= shimNumGroups) throw OpenMMException("Index out of range");
    shimParticleGroup.push_back(tempGroup);
    return (int)shimParticleGroup.size() - 1;
}
double DrudeTGNHIntegrator::getTemperature() const { return shimT; }
double DrudeTGNHIntegrator::getCouplingTime() const { return shimTau; }
double DrudeTGNHIntegrator::getDrudeTemperature() const { return shimDrudeT; }
double DrudeTGNHIntegrator::getDrudeCouplingTime() const { return shimDrudeTau; }
double DrudeTGNHIntegrator::getStepSize() const { return shimDt; }
double DrudeTGNHIntegrator::getConstraintTolerance() const { return shimTol; }
double DrudeTGNHIntegrator::getMaxDrudeDistance() const { return shimMaxDist; }
int DrudeTGNHIntegrator::getDrudeStepsPerRealStep() const { return shimDrudeSteps; }
int DrudeTGNHIntegrator::getNumNHChains() const { return shimChains; }
bool DrudeTGNHIntegrator::getUseDrudeNHChains() const { return shimDrudeChains; }
bool DrudeTGNHIntegrator::getUseCOMTempGroup() const { return shimCOM; }
int DrudeTGNHIntegrator::getNumTempGroups() const { return shimNumGroups; }
int DrudeTGNHIntegrator::getNumResidues() const { return shimNumResidues; }
int DrudeTGNHIntegrator::getParticleResId(int particle) const { return shimResId.at(particle); }
void DrudeTGNHIntegrator::getParticleTempGroup(int particle, int& tempGroup) const {
    if (particle < 0 || particle >= (int)shimParticleGroup.size()) throw OpenMMException("Index out of range");
    tempGroup = shimParticleGroup[particle];
}
bool DrudeTGNHIntegrator::isKineticEnergySumValid() const { return shimKEValid; }

KernelImpl::KernelImpl(std::string name, const Platform&) : shimName(name) {}
KernelImpl::~KernelImpl() {}
std::string IntegrateDrudeTGNHStepKernel::Name() { return "IntegrateDrudeTGNHStep"; }
IntegrateDrudeTGNHStepKernel::IntegrateDrudeTGNHStepKernel(std::string name, const Platform& platform) : KernelImpl(name, platform) {}
KernelFactory::~KernelFactory() {}

static std::map<std::string, Platform*>& registry() { static std::map<std::string, Platform*> r; return r; }
Platform::~Platform() {}
Platform& Platform::getPlatformByName(const std::string& name) {
    std::map<std::string, Platform*>::iterator it = registry().find(name);
    if (it == registry().end()) throw OpenMMException("There is no registered Platform called \"" + name + "\"");
    return *it->second;
}
void Platform::registerPlatform(Platform* platform) { registry()[platform->shimPlatformName] = platform; }
void Platform::registerKernelFactory(const std::string& name, KernelFactory* factory) { shimFactories[name] = factory; }

double ContextImpl::calcForcesAndEnergy(bool includeForces, bool, int) {
    shimForceCalls++;
    if (includeForces && shimForces) shimForces();
    return 0.0;
}
void* ContextImpl::getPlatformData() { return shimPlatformData; }

void* HipArray::getDevicePointer() { return shimPtr; }
HipArray& IntegrationUtilities::getPosDelta() { return shimPosDelta; }
void IntegrationUtilities::applyConstraints(double tol) { if (shimApplyConstraints) shimApplyConstraints(tol); }
void IntegrationUtilities::applyVelocityConstraints(double tol) { if (shimApplyVelocityConstraints) shimApplyVelocityConstraints(tol); }
void IntegrationUtilities::computeVirtualSites() { if (shimVirtualSites) shimVirtualSites(); }
double IntegrationUtilities::computeKineticEnergy(double timeShift) { return shimKineticEnergy ? shimKineticEnergy(timeShift) : 0.0; }

void HipPlatform::PlatformData::initializeContexts(const System&) {}
ComputeContext::~ComputeContext() {}
HipPlatform::PlatformData& HipContext::getPlatformData() { return *shimData; }
bool HipContext::getUseDoublePrecision() const { return shimDouble; }
bool HipContext::getUseMixedPrecision() const { return shimMixed; }
int HipContext::getDeviceIndex() const { return shimDevice; }
int HipContext::getPaddedNumAtoms() const { return shimPadded; }
IntegrationUtilities& HipContext::getIntegrationUtilities() { return shimUtilities; }
void* HipContext::getCurrentStream() { return shimStream; }
HipArray& HipContext::getPosq() { return shimPosq; }
HipArray& HipContext::getPosqCorrection() { return shimPosqCorrection; }
HipArray& HipContext::getVelm() { return shimVelm; }
HipArray& HipContext::getForce() { return shimForce; }
bool HipContext::getAtomsWereReordered() const { return false; }      // (this runtime never reorders: the test's arrays stay in input order)
void HipContext::reorderAtoms() { shimReorders++; }
double HipContext::getTime() { return shimTime; }
void HipContext::setTime(double t) { shimTime = t; }
long long HipContext::getStepCount() { return shimStepCount; }
void HipContext::setStepCount(long long n) { shimStepCount = n; }
ContextSelector::ContextSelector(ComputeContext&) {}
ContextSelector::~ContextSelector() {}

// serialization: a tree of named nodes with string properties (numbers as text with 17 significant digits, so that a double
// comes back as it went in)
const std::string& SerializationNode::getName() const { return shimNodeName; }
const std::vector<SerializationNode>& SerializationNode::getChildren() const { return shimChildren; }
SerializationNode& SerializationNode::createChildNode(const std::string& name) {
    shimChildren.emplace_back();
    shimChildren.back().shimNodeName = name;
    return shimChildren.back();
}
bool SerializationNode::hasProperty(const std::string& name) const { return shimProperties.count(name) != 0; }
static const std::string& property(const SerializationNode& n, const std::string& name) {
    std::map<std::string, std::string>::const_iterator it = n.shimProperties.find(name);
    if (it == n.shimProperties.end()) throw OpenMMException("Unknown property '" + name + "' in node '" + n.shimNodeName + "'");
    return it->second;
}
SerializationNode& SerializationNode::setIntProperty(const std::string& name, int value) { shimProperties[name] = std::to_string(value); return *this; }
int SerializationNode::getIntProperty(const std::string& name) const { return std::atoi(property(*this, name).c_str()); }
SerializationNode& SerializationNode::setDoubleProperty(const std::string& name, double value) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.17g", value);
    shimProperties[name] = buf;
    return *this;
}
double SerializationNode::getDoubleProperty(const std::string& name) const { return std::strtod(property(*this, name).c_str(), nullptr); }
SerializationNode& SerializationNode::setStringProperty(const std::string& name, const std::string& value) { shimProperties[name] = value; return *this; }
const std::string& SerializationNode::getStringProperty(const std::string& name) const { return property(*this, name); }
bool SerializationNode::getBoolProperty(const std::string& name) const { return std::atoi(property(*this, name).c_str()) != 0; }

static std::map<std::string, const SerializationProxy*>& proxies() { static std::map<std::string, const SerializationProxy*> p; return p; }
SerializationProxy::SerializationProxy(const std::string& typeName) : shimTypeName(typeName) {}
SerializationProxy::~SerializationProxy() {}
void SerializationProxy::registerProxy(const std::type_info& type, const SerializationProxy* proxy) { proxies()[type.name()] = proxy; }
const SerializationProxy* shimFindProxy(const std::type_info& type) {
    std::map<std::string, const SerializationProxy*>::iterator it = proxies().find(type.name());
    return it == proxies().end() ? nullptr : it->second;
}

}  // namespace OpenMM
