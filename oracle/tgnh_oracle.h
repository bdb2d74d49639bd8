/*
 * tgnh_oracle.h -- CPU oracle for the DrudeTGNHIntegrator per-timestep path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PIN STATUS: STATISTICAL ONLY.  The reference (scychon/openmm_drudeNose) holds no
 * golden vectors or bit-level known-answer values for this path, and its sources
 * cannot be built in this image without OpenMM (absent), so there are no reference
 * outputs to compare against: at the level of individual trajectories parity is
 * UNPINNED.  What the reference's tests do hold is checked: its one enabled test,
 * testWater (mean temperature of 216 rigid SWM4 waters within 3 % of the
 * dof-weighted target), passes on this oracle (tests/test_reference_water.py,
 * +0.7 %), with the test's force field restated in water_ff.c; of its disabled
 * testSinglePair two of three assertions hold (tests/test_oracle.py).
 * This file is a plain-C restatement of the reference algorithm, written from the
 * reference text, each function citing the file:line it follows.  See DESIGN.md 6.
 *
 * Two semantic modes:
 *   TGO_MODE_DUALNH  follows platforms/reference/src/ReferenceDrudeTGNHKernels.cpp
 *                    (dual real/Drude Nose-Hoover chain, bug-compatible indexing)
 *   TGO_MODE_TGNH    follows platforms/cuda/src/CudaDrudeTGNHKernels.cpp and
 *                    platforms/cuda/src/kernels/drudeTGNH.cu (per-temperature-group
 *                    + molecular-COM + Drude thermostats)
 * All arithmetic is double precision, sums run in index order.
 */
#ifndef TGNH_ORACLE_H_
#define TGNH_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

enum { TGO_MODE_DUALNH = 0, TGO_MODE_TGNH = 1 };

/* status codes: 0 ok, <0 error; tgo_last_error() has the text. */
enum {
    TGO_OK = 0,
    TGO_ERR_ARG = -1,
    TGO_ERR_GROUP_MISMATCH = -2,     /* CudaDrudeTGNHKernels.cpp:145-146, 192-193 */
    TGO_ERR_HARDWALL = -3            /* ReferenceDrudeTGNHKernels.cpp:311-312 */
};

typedef struct tgo_desc {
    int mode;                    /* TGO_MODE_* */
    int num_particles;           /* N */
    int num_pairs;               /* P, in DrudeForce order */
    int num_groups;              /* G = getNumTempGroups() (TGNH); ignored by DUALNH */
    int num_residues;            /* R (TGNH) */
    int num_constraints;         /* count only: the dof bookkeeping */
    int has_cm_motion_remover;   /* typeid(CMMotionRemover) sniff */
    const double* mass;          /* [N] */
    const int* pair_drude;       /* [P]  particles.x / pair.first  (p)  */
    const int* pair_parent;      /* [P]  particles.y / pair.second (p1) */
    const int* group;            /* [N]  getParticleTempGroup (TGNH) */
    const int* resid;            /* [N]  getParticleResId (TGNH) */
    const int* constraint_i;     /* [num_constraints] or NULL (TGNH group check) */
    const int* constraint_j;
    double kB;                   /* BOLTZ, kJ/mol/K (explicit: OpenMM version unpinned) */
    double temperature, coupling_time;
    double drude_temperature, drude_coupling_time;
    double step_size;
    int drude_steps_per_real_step;   /* S */
    int num_nh_chains;               /* C */
    int use_drude_nh_chains;
    int use_com_temp_group;
    double max_drude_distance;       /* 0 = hard wall off */
} tgo_desc;

typedef struct tgo_state tgo_state;

const char* tgo_last_error(void);

int  tgo_create(const tgo_desc* d, tgo_state** out);
void tgo_destroy(tgo_state* s);

/* Integrator scalars the reference re-reads every step. */
void tgo_set_step_size(tgo_state* s, double dt);
void tgo_set_drude_steps(tgo_state* s, int n);
void tgo_set_max_drude_distance(tgo_state* s, double d);

/* A1: topology.  normal = ascending indices of particles in no pair. */
int  tgo_num_normal(const tgo_state* s);
void tgo_get_normal(const tgo_state* s, int* out);
/* A2: dof / NkT / Q.  DUALNH: n = 2 (real, drude).  TGNH: n = G+2. */
int  tgo_num_thermostats(const tgo_state* s);
void tgo_get_dof(const tgo_state* s, double* dof /*[n]*/, double* nkt /*[n]*/);
/* Thermostat arrays, flat.  DUALNH: the reference's interleaved vectors
 * (sizes via tgo_chain_len).  TGNH: [itg][link] with etaDot rows of C+1. */
int  tgo_chain_len(const tgo_state* s, int which /*0 eta 1 etaDot 2 etaDotDot 3 etaMass*/);
void tgo_get_chain(const tgo_state* s, int which, double* out);
void tgo_set_chain(tgo_state* s, int which, const double* in);

/* A3/A4 only: kinetic energies (no 1/2), n values as tgo_num_thermostats. */
void tgo_kinetic_energies(const tgo_state* s, const double* vel /*[N][3]*/, double* ke);

/* A3..A6: KE -> chain -> rescale velocities.  ke_out = KE before the chain,
 * scale_out = velocity scale factors applied; either may be NULL. */
void tgo_propagate_nhc(tgo_state* s, double* vel, double* ke_out, double* scale_out);
/* A5 alone: chain on given kinetic energies, returns scale factors. */
void tgo_chain_only(tgo_state* s, const double* ke_in, double* scale_out);
/* A6 alone */
void tgo_scale_velocities(const tgo_state* s, double* vel, const double* scale);
/* A7 */
void tgo_half_kick(const tgo_state* s, double* vel, const double* force);
/* A8 (no constraints): x' = x + v dt ; v = (x'-x)/dt ; x = x' (massive only) */
void tgo_drift(const tgo_state* s, double* pos, double* vel);
/* A10; returns TGO_ERR_HARDWALL in DUALNH mode when r > 2 max. */
int  tgo_hardwall(const tgo_state* s, double* pos, double* vel);

/* A11 split around the force call-out.  begin = nhc, kick, drift, hard wall.
 * end = kick, nhc, time += dt. */
int  tgo_step_begin(tgo_state* s, double* pos, double* vel, const double* force);
int  tgo_step_end(tgo_state* s, double* vel, const double* force);

/* A12.  TGNH: cached KESum (valid) or 1/2 sum m v^2.  DUALNH: half-step
 * shifted KE (no constraints): 1/2 sum m (v + F dt/2 /m)^2. */
double tgo_kinetic_energy_query(const tgo_state* s, const double* vel, const double* force, int ke_sum_valid);

/* Harness force (test/bench workload, not part of the reference): Drude spring
 * k_drude between pair members + tether k_tether of every massive non-Drude
 * particle to its site x0.  Massless slots get zero force. */
void tgo_harness_force(const tgo_state* s, const double* pos, const double* x0,
                       double k_drude, double k_tether, double* force);
/* Whole loop with the harness force, for the timed CPU baseline and 100-step
 * parity runs.  force must hold F(pos) on entry; holds F(final pos) on exit. */
int  tgo_run_harness(tgo_state* s, double* pos, double* vel, double* force, const double* x0,
                     double k_drude, double k_tether, int nsteps);

/* ---- harness call-outs for the constrained path (NOT from the reference, which delegates both to OpenMM:
 * Ref :268 ReferenceConstraints::apply ; Cu :363 applyConstraints, :391 applyVelocityConstraints, Ref :373 /
 * Cu :377 virtual sites).  Constraint clusters: up to 4 atoms with up to 6 distance constraints among them,
 * solved by SHAKE sweeps in cluster order until every |d^2 - r^2| <= 2 tol d^2. ---- */
void tgo_set_clusters(tgo_state* s, int nclusters, const int* atoms /*[n][4], -1 = unused*/,
                      const int* ncons /*[n]*/, const int* pairs /*[n][6][2] local indices*/, const double* dist /*[n][6]*/);
/* positions: old positions `pos`, proposed displacement `delta` (x' = pos + delta), corrected in place */
int  tgo_shake_positions(const tgo_state* s, const double* pos, double* delta, double tol);
/* velocities: remove the components along the constrained bonds (RATTLE velocity stage) */
int  tgo_shake_velocities(const tgo_state* s, const double* pos, double* vel, double tol);
/* three-particle-average virtual sites: pos[site] = w1 pos[p1] + w2 pos[p2] + w3 pos[p3] */
void tgo_set_virtual_sites(tgo_state* s, int n, const int* atoms /*[n][4] site,p1,p2,p3*/, const double* w /*[n][3]*/);
void tgo_virtual_sites(const tgo_state* s, double* pos);
/* A11 with the call-outs in place.  DUALNH (Ref :253-284): x' = x + v dt, constrain, v = (x'-x)/dt, no velocity
 * stage.  TGNH (Cu :356-391): posDelta = dt v, constrain, x += posDelta, v = posDelta/dt, velocity stage after
 * the second kick. */
int  tgo_run_harness_constrained(tgo_state* s, double* pos, double* vel, double* force, const double* x0,
                                 double k_drude, double k_tether, double tol, int nsteps);

double tgo_time(const tgo_state* s);
long   tgo_step_count(const tgo_state* s);

#ifdef __cplusplus
}
#endif
#endif
