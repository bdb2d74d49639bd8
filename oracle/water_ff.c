/*
 * water_ff.c -- the force field of the reference's testWater, for the one test of the reference that is enabled
 * (platforms/reference/tests/TestReferenceDrudeTGNHIntegrator.cpp:111-192, main() :257-259).
 *
 * TEST INFRASTRUCTURE ONLY (part of the oracle).  The reference gets these forces from OpenMM; OpenMM is absent
 * here, so this file restates what that test asks OpenMM for, from OpenMM's published definitions:
 *   - NonbondedForce, CutoffPeriodic, cutoff 1.0 nm (test :126-127): Coulomb with reaction field
 *     (eps_rf = 78.3, OpenMM's default), E = k q1 q2 (1/r + krf r^2 - crf), and Lennard-Jones
 *     4 eps ((s/r)^12 - (s/r)^6) with Lorentz-Berthelot combination; minimum image per atom pair;
 *     every intramolecular pair is an exception with zero charge product and zero epsilon (test :140-142).
 *   - per molecule O, D, H1, H2, M (test :130-139): charges 1.71636, -1.71636, 0.55733, 0.55733, -1.11466;
 *     only O has LJ (sigma 0.318395 nm, eps 0.21094*4.184 kJ/mol).
 *   - DrudeForce (test :148): isotropic spring between D and O, k = ONE_4PI_EPS0 q^2 / alpha with
 *     alpha = ONE_4PI_EPS0*1.71636^2/(100000*4.184)  =>  k = 418400 kJ/mol/nm^2.
 *   - M is a ThreeParticleAverageSite of (O, H1, H2) (test :147): its force is spread over them by the weights.
 * Velocity-independent, deterministic; forces in kJ/mol/nm.
 */
#include <math.h>
#include <string.h>

#define ONE_4PI_EPS0 138.935456

static const double Q[5] = {1.71636, -1.71636, 0.55733, 0.55733, -1.11466};
static const double W[3] = {0.786646558, 0.106676721, 0.106676721};

double tgo_water_forces(int n_mol, const double* pos, double box, double cutoff, double* force) {
    const int n = 5 * n_mol;
    const double eps_rf = 78.3;
    const double krf = (1.0 / (cutoff * cutoff * cutoff)) * (eps_rf - 1.0) / (2.0 * eps_rf + 1.0);
    const double crf = (1.0 / cutoff) * (3.0 * eps_rf) / (2.0 * eps_rf + 1.0);
    const double sigma = 0.318395, eps = 0.21094 * 4.184;
    const double kd = 100000.0 * 4.184;
    const double c2 = cutoff * cutoff, inv_box = 1.0 / box;
    double energy = 0.0;
    memset(force, 0, sizeof(double) * 3 * (size_t)n);
    for (int a = 0; a < n_mol; a++) {
        for (int b = a + 1; b < n_mol; b++) {
            for (int i = 0; i < 5; i++) {
                const double* pi = pos + 3 * (5 * a + i);
                for (int j = 0; j < 5; j++) {
                    const double* pj = pos + 3 * (5 * b + j);
                    double d[3];
                    for (int k = 0; k < 3; k++) {
                        d[k] = pi[k] - pj[k];
                        d[k] -= box * floor(d[k] * inv_box + 0.5);
                    }
                    const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                    if (r2 >= c2) continue;
                    const double r = sqrt(r2), inv_r = 1.0 / r;
                    const double qq = ONE_4PI_EPS0 * Q[i] * Q[j];
                    double dEdr = qq * (-inv_r * inv_r + 2.0 * krf * r);      /* dE/dr */
                    energy += qq * (inv_r + krf * r2 - crf);
                    if (i == 0 && j == 0) {
                        const double s2 = sigma * sigma / r2, s6 = s2 * s2 * s2;
                        energy += 4.0 * eps * (s6 * s6 - s6);
                        dEdr += 4.0 * eps * (-12.0 * s6 * s6 + 6.0 * s6) * inv_r;
                    }
                    for (int k = 0; k < 3; k++) {
                        const double f = -dEdr * d[k] * inv_r;
                        force[3 * (5 * a + i) + k] += f;
                        force[3 * (5 * b + j) + k] -= f;
                    }
                }
            }
        }
    }
    for (int a = 0; a < n_mol; a++) {
        double* fo = force + 3 * (5 * a), *fd = fo + 3, *fh1 = fo + 6, *fh2 = fo + 9, *fm = fo + 12;
        const double* po = pos + 3 * (5 * a), *pd = po + 3;
        for (int k = 0; k < 3; k++) {
            const double s = pd[k] - po[k];                   /* Drude spring */
            energy += 0.5 * kd * s * s;
            fd[k] -= kd * s;
            fo[k] += kd * s;
            fo[k] += W[0] * fm[k]; fh1[k] += W[1] * fm[k]; fh2[k] += W[2] * fm[k];   /* virtual-site force */
            fm[k] = 0.0;
        }
    }
    return energy;
}
