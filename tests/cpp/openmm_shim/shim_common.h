// NOT OpenMM: stand-ins (see README.md in this directory).  One header holds them all; the files named after OpenMM's headers only
// include it.  Two uses: declarations only (tests/test_glue_syntax.py: g++ -fsyntax-only over the glue), and -- with
// -DTGNH_SHIM_FUNCTIONAL, which gives the classes the few data members below, and shim_runtime.cpp, which gives the functions
// bodies -- a miniature runtime behind the same signatures that tests/cpp/test_glue_linked.cpp links the REAL glue sources against
// (tests/test_glue_linked_gpu.py).  SHIM_STATE(...) members exist in the functional build only.
#ifndef TGNH_OPENMM_SHIM_H_
#define TGNH_OPENMM_SHIM_H_
#include <exception>
#include <typeinfo>
#include <string>
#include <vector>
#ifdef TGNH_SHIM_EXPORT_DEFAULT
#define OPENMM_EXPORT __attribute__((visibility("default")))       // what OpenMM's windowsExport.h gives a Linux build
#else
#define OPENMM_EXPORT
#endif
#define BOLTZ 8.31446261815324e-3
#ifdef TGNH_SHIM_FUNCTIONAL
#include <functional>
#include <map>
#define SHIM_STATE(...) public: __VA_ARGS__
#else
#define SHIM_STATE(...)
#endif
namespace OpenMM {
class System; class Platform; class ContextImpl; class HipContext; class DrudeForce; class DrudeTGNHIntegrator;
class OpenMMException : public std::exception {
public:
    explicit OpenMMException(const std::string& message);
    const char* what() const noexcept override;
    SHIM_STATE(std::string shimMessage;)
};
class Force { public: virtual ~Force(); };
class CMMotionRemover : public Force {};
class AndersenThermostat : public Force {};
class System {
public:
    int getNumParticles() const;
    double getParticleMass(int index) const;
    int getNumConstraints() const;
    void getConstraintParameters(int index, int& particle1, int& particle2, double& distance) const;
    int getNumForces() const;
    const Force& getForce(int index) const;
    SHIM_STATE(struct ShimConstraint { int a, b; double d; };
               std::vector<double> shimMass; std::vector<ShimConstraint> shimConstraints; std::vector<const Force*> shimForces;)
};
class DrudeForce : public Force {
public:
    int getNumParticles() const;
    void getParticleParameters(int index, int& particle, int& particle1, int& particle2, int& particle3, int& particle4,
                               double& charge, double& polarizability, double& aniso12, double& aniso34) const;
    SHIM_STATE(std::vector<int> shimDrude, shimParent;)
};
class DrudeTGNHIntegrator {
public:
    DrudeTGNHIntegrator(double temperature, double couplingTime, double drudeTemperature, double drudeCouplingTime, double stepSize,
                        int drudeStepsPerRealStep = 20, int numNHChains = 1, bool useDrudeNHChains = false, bool useCOMTempGroup = true);
    void setConstraintTolerance(double tol); void setMaxDrudeDistance(double distance); void setUseCOMTempGroup(int useCOMGroup);
    int addTempGroup(); int addParticleTempGroup(int tempGroup);
    double getTemperature() const; double getCouplingTime() const; double getDrudeTemperature() const; double getDrudeCouplingTime() const;
    double getStepSize() const; double getConstraintTolerance() const; double getMaxDrudeDistance() const;
    int getDrudeStepsPerRealStep() const; int getNumNHChains() const; bool getUseDrudeNHChains() const; bool getUseCOMTempGroup() const;
    int getNumTempGroups() const; int getNumResidues() const; int getParticleResId(int particle) const;
    void getParticleTempGroup(int particle, int& tempGroup) const;
    bool isKineticEnergySumValid() const;      // the accessor -DDRUDETGNH_TRUST_STATE_CHANGED asks the API class for
    SHIM_STATE(double shimT, shimTau, shimDrudeT, shimDrudeTau, shimDt, shimTol = 1e-5, shimMaxDist = 0;
               int shimDrudeSteps, shimChains; bool shimDrudeChains, shimCOM, shimKEValid = false;
               int shimNumGroups = 0; std::vector<int> shimParticleGroup, shimResId; int shimNumResidues = 0;)   // (the reference fills residues from Context::getMolecules, API :103-160: the test does)
};
class KernelImpl {
public:
    KernelImpl(std::string name, const Platform& platform);
    virtual ~KernelImpl();
    SHIM_STATE(std::string shimName;)
};
class IntegrateDrudeTGNHStepKernel : public KernelImpl {
public:
    static std::string Name();
    IntegrateDrudeTGNHStepKernel(std::string name, const Platform& platform);
    virtual void initialize(const System& system, const DrudeTGNHIntegrator& integrator, const DrudeForce& force) = 0;
    virtual void execute(ContextImpl& context, const DrudeTGNHIntegrator& integrator) = 0;
    virtual double computeKineticEnergy(ContextImpl& context, const DrudeTGNHIntegrator& integrator, bool isKESumValid) = 0;
};
class KernelFactory {
public:
    virtual ~KernelFactory();
    virtual KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const = 0;
};
class Platform {
public:
    virtual ~Platform();
    static Platform& getPlatformByName(const std::string& name);
    static void registerPlatform(Platform* platform);
    void registerKernelFactory(const std::string& name, KernelFactory* factory);
    SHIM_STATE(std::string shimPlatformName = "Reference"; std::map<std::string, KernelFactory*> shimFactories;)
};
class ContextImpl {
public:
    double calcForcesAndEnergy(bool includeForces, bool includeEnergy, int groups = -1);
    void* getPlatformData();
    SHIM_STATE(std::function<void()> shimForces; void* shimPlatformData = nullptr; int shimForceCalls = 0;)
};
class HipArray { public: void* getDevicePointer(); SHIM_STATE(void* shimPtr = nullptr;) };
class IntegrationUtilities {
public:
    HipArray& getPosDelta();
    void applyConstraints(double tol); void applyVelocityConstraints(double tol); void computeVirtualSites();
    double computeKineticEnergy(double timeShift);
    SHIM_STATE(HipArray shimPosDelta; std::function<void(double)> shimApplyConstraints, shimApplyVelocityConstraints;
               std::function<void()> shimVirtualSites; std::function<double(double)> shimKineticEnergy;)
};
class HipPlatform : public Platform {
public:
#ifdef TGNH_SHIM_FUNCTIONAL
    HipPlatform() { shimPlatformName = "HIP"; }
#endif
    class PlatformData {
    public:
        void initializeContexts(const System& system);
        std::vector<HipContext*> contexts;
    };
};
class ComputeContext { public: virtual ~ComputeContext(); };
class HipContext : public ComputeContext {
public:
    HipPlatform::PlatformData& getPlatformData();
    bool getUseDoublePrecision() const; bool getUseMixedPrecision() const; int getDeviceIndex() const; int getPaddedNumAtoms() const;
    IntegrationUtilities& getIntegrationUtilities();
    void* getCurrentStream();
    HipArray& getPosq(); HipArray& getPosqCorrection(); HipArray& getVelm(); HipArray& getForce();
    bool getAtomsWereReordered() const; void reorderAtoms();
    double getTime(); void setTime(double t); long long getStepCount(); void setStepCount(long long n);
    SHIM_STATE(HipPlatform::PlatformData* shimData = nullptr; bool shimDouble = false, shimMixed = true; int shimDevice = 0, shimPadded = 0;
               IntegrationUtilities shimUtilities; void* shimStream = nullptr; HipArray shimPosq, shimPosqCorrection, shimVelm, shimForce;
               double shimTime = 0; long long shimStepCount = 0; int shimReorders = 0;)
};
class ContextSelector { public: explicit ContextSelector(ComputeContext& context); ~ContextSelector(); };
class SerializationNode {
public:
    const std::string& getName() const;
    const std::vector<SerializationNode>& getChildren() const;
    SerializationNode& createChildNode(const std::string& name);
    bool hasProperty(const std::string& name) const;
    SerializationNode& setIntProperty(const std::string& name, int value); int getIntProperty(const std::string& name) const;
    SerializationNode& setDoubleProperty(const std::string& name, double value); double getDoubleProperty(const std::string& name) const;
    SerializationNode& setStringProperty(const std::string& name, const std::string& value); const std::string& getStringProperty(const std::string& name) const;
    bool getBoolProperty(const std::string& name) const;
    SHIM_STATE(std::string shimNodeName; std::map<std::string, std::string> shimProperties; std::vector<SerializationNode> shimChildren;)
};
class SerializationProxy {
public:
    explicit SerializationProxy(const std::string& typeName);
    virtual ~SerializationProxy();
    virtual void serialize(const void* object, SerializationNode& node) const = 0;
    virtual void* deserialize(const SerializationNode& node) const = 0;
    static void registerProxy(const std::type_info& type, const SerializationProxy* proxy);
    SHIM_STATE(std::string shimTypeName;)
};
}  // namespace OpenMM
#endif
