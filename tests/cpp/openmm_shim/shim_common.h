// NOT OpenMM: declarations-only stand-ins (see README.md in this directory).  One header holds them all; the files named after
// OpenMM's headers only include it.
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
namespace OpenMM {
class System; class Platform; class ContextImpl; class HipContext; class DrudeForce; class DrudeTGNHIntegrator;
class OpenMMException : public std::exception {
public:
    explicit OpenMMException(const std::string& message);
    const char* what() const noexcept override;
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
};
class DrudeForce : public Force {
public:
    int getNumParticles() const;
    void getParticleParameters(int index, int& particle, int& particle1, int& particle2, int& particle3, int& particle4,
                               double& charge, double& polarizability, double& aniso12, double& aniso34) const;
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
};
class KernelImpl {
public:
    KernelImpl(std::string name, const Platform& platform);
    virtual ~KernelImpl();
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
};
class ContextImpl {
public:
    double calcForcesAndEnergy(bool includeForces, bool includeEnergy, int groups = -1);
    void* getPlatformData();
};
class HipArray { public: void* getDevicePointer(); };
class IntegrationUtilities {
public:
    HipArray& getPosDelta();
    void applyConstraints(double tol); void applyVelocityConstraints(double tol); void computeVirtualSites();
    double computeKineticEnergy(double timeShift);
};
class HipPlatform : public Platform {
public:
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
};
class SerializationProxy {
public:
    explicit SerializationProxy(const std::string& typeName);
    virtual ~SerializationProxy();
    virtual void serialize(const void* object, SerializationNode& node) const = 0;
    virtual void* deserialize(const SerializationNode& node) const = 0;
    static void registerProxy(const std::type_info& type, const SerializationProxy* proxy);
};
}  // namespace OpenMM
#endif
