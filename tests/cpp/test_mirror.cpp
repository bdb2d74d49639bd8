// C++ check of include/DrudeTGNHIntegratorHip.hpp through a host-only handle (no GPU): API behaviour of the
// reference class (openmmapi/src/DrudeTGNHIntegrator.cpp) and the dof the library derives.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "DrudeTGNHIntegratorHip.hpp"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
template <class F> static bool throws(F&& f) { try { f(); } catch (const std::exception&) { return true; } return false; }

int main() {
    using namespace drudetgnh;
    DrudeTGNHIntegrator integ(300.0, 0.1, 1.0, 0.005, 0.001);
    REQUIRE(integ.getDrudeStepsPerRealStep() == 20 && integ.getNumNHChains() == 1);
    REQUIRE(integ.getUseDrudeNHChains() == 0 && integ.getUseCOMTempGroup());          // C++ defaults (header :71)
    REQUIRE(integ.getConstraintTolerance() == 1e-5 && integ.getMaxDrudeDistance() == 0);
    REQUIRE(throws([&] { integ.setMaxDrudeDistance(-1); }));
    REQUIRE(throws([&] { integ.addParticleTempGroup(0); }));
    REQUIRE(throws([&] { integ.computeKineticEnergy(nullptr); }));                    // not bound to a context
    // 8 SWM4 waters: O, D, H, H, M
    SystemDesc sys;
    const double m[5] = {15.6, 0.4, 1.0, 1.0, 0.0};
    for (int w = 0; w < 8; w++) {
        std::vector<int> mol;
        for (int k = 0; k < 5; k++) { sys.mass.push_back(m[k]); mol.push_back(5 * w + k); }
        sys.molecules.push_back(mol);
        sys.drudePairs.push_back({5 * w + 1, 5 * w});
    }
    sys.hasCMMotionRemover = true;
    integ.initialize(sys, -1);
    REQUIRE(integ.getNumTempGroups() == 1 && integ.getNumResidues() == 8);
    REQUIRE(std::fabs(integ.getResInvMass(3) - 1.0 / 18.0) < 1e-15 && integ.getParticleResId(12) == 2);
    int nt = 0;
    REQUIRE(tgnh_get_num_thermostats(integ.getHandle(), &nt) == TGNH_OK && nt == 3);
    double dof[3], nkt[3];
    REQUIRE(tgnh_get_dof(integ.getHandle(), dof, nkt) == TGNH_OK);
    REQUIRE(dof[0] == 3 * 32 - 3 * 8 - 3 * 8 && dof[1] == 3 * 8 - 3 && dof[2] == 3 * 8);   // Cu :129-132, :148, :198, :208
    REQUIRE(tgnh_step_begin(integ.getHandle(), nullptr) == TGNH_ERR_STATE);          // host-only handle cannot launch
    DrudeTGNHIntegrator bad(300.0, 0.1, 1.0, 0.005, 0.001);
    bad.addTempGroup(); bad.addTempGroup();
    for (int i = 0; i < 40; i++) bad.addParticleTempGroup(i == 1 ? 1 : 0);            // Drude of water 0 in another group
    REQUIRE(throws([&] { bad.initialize(sys, -1); }));                                // Cu :145-146
    std::printf("OK\n");
    return 0;
}
