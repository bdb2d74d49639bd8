/* SWIG interface of the Python module `drudetgnhplugin` for an OpenMM (>= 8) installation with the HIP platform.
 *
 * The surface is the reference's (python/drudetgnhplugin.i:60-92 of scychon/openmm_drudeNose), because user scripts
 * such as example/nacl_tg.py are written against it: module name, class OpenMM::DrudeTGNHIntegrator, the nine
 * constructor arguments with the Python-side default useDrudeNHChains=True (:62; the C++ default is false), the 24
 * methods, getParticleTempGroup returning its result (int& OUTPUT, :88-90), unit-carrying getters (:35-53).  The class
 * it wraps is the reference plugin's own platform-independent API class (openmmapi/, libOpenMMDrudeTGNH), which stays as
 * it is; this repository adds the HIP platform's kernel underneath it (platforms/hip).
 *
 * Not buildable in this repository's image (no SWIG, no OpenMM); openmm_drudenose_amd/drudetgnhplugin.py mirrors the same
 * surface over the C ABI for the tests. */
%module drudetgnhplugin

%import(module="openmm") "swig/OpenMMSwigHeaders.i"
%include "swig/typemaps.i"
%include "std_vector.i"
namespace std {
    %template(vectord) vector<double>;
    %template(vectori) vector<int>;
}

%{
#include "OpenMM.h"
#include "OpenMMDrude.h"
#include "OpenMMDrudeTGNH.h"
%}

%pythoncode %{
import openmm as mm
import openmm.unit as unit
%}

/* getters that hand back a quantity with its unit */
%define TGNH_UNIT(METHOD, UNIT)
%pythonappend OpenMM::DrudeTGNHIntegrator::METHOD() const %{
    val = unit.Quantity(val, UNIT)
%}
%enddef
TGNH_UNIT(getTemperature, unit.kelvin)
TGNH_UNIT(getDrudeTemperature, unit.kelvin)
TGNH_UNIT(getCouplingTime, unit.picosecond)
TGNH_UNIT(getDrudeCouplingTime, unit.picosecond)
TGNH_UNIT(getMaxDrudeDistance, unit.nanometer)

namespace OpenMM {

class DrudeTGNHIntegrator : public Integrator {
public:
    DrudeTGNHIntegrator(double temperature, double couplingTime, double drudeTemperature, double drudeCouplingTime,
                        double stepSize, int drudeStepsPerRealStep = 20, int numNHChains = 1,
                        int useDrudeNHChains = True, int useCOMTempGroup = True);

    /* the temperature-group table (user scripts fill it before the Context is made) */
    int addTempGroup();
    int addParticleTempGroup(int tempGroup);
    void setParticleTempGroup(int particle, int tempGroup);
    %apply int& OUTPUT { int& tempGroup };
    void getParticleTempGroup(int particle, int& tempGroup) const;
    %clear int& tempGroup;
    int getNumTempGroups() const;

    /* what the thermostats are asked for: read ... */
    double getTemperature() const;
    double getDrudeTemperature() const;
    double getCouplingTime() const;
    double getDrudeCouplingTime() const;
    double getMaxDrudeDistance() const;
    int getNumNHChains() const;
    int getDrudeStepsPerRealStep() const;
    int getUseDrudeNHChains() const;
    int getUseCOMTempGroup() const;

    /* ... and write (temperatures in kelvin, times in picoseconds, the hard-wall distance in nanometres; 0 = no wall) */
    void setTemperature(double kelvin);
    void setDrudeTemperature(double kelvin);
    void setCouplingTime(double picoseconds);
    void setDrudeCouplingTime(double picoseconds);
    void setMaxDrudeDistance(double nanometres);
    void setNumNHChains(int links);
    void setDrudeStepsPerRealStep(int substeps);
    void setUseDrudeNHChains(int yes);
    void setUseCOMTempGroup(int yes);

    virtual void step(int steps);
};

}
