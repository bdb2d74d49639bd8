/*
 * tgnh_oracle.c -- CPU oracle for the DrudeTGNHIntegrator per-timestep path.
 *
 * TEST INFRASTRUCTURE ONLY (see tgnh_oracle.h).  Pin status: statistical only (ibid.).
 *
 * Citations: "Ref" = platforms/reference/src/ReferenceDrudeTGNHKernels.cpp,
 *            "Cu"  = platforms/cuda/src/CudaDrudeTGNHKernels.cpp,
 *            "K"   = platforms/cuda/src/kernels/drudeTGNH.cu,
 *            "API" = openmmapi/src/DrudeTGNHIntegrator.cpp
 * of scychon/openmm_drudeNose.  Compile with -ffp-contract=off so that the
 * operation order below is the operation order executed.
 */
#include "tgnh_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Deliberate mis-restatements, for tests/pin_sensitivity.py only ("which of the reference's own checks would notice?").
 * They exist in libtgnh_oracle_mut.so (-DTGO_MUTANTS); in the oracle proper MUT(k) is the constant 0. */
#ifdef TGO_MUTANTS
static int g_mutant = 0;
void tgo_set_mutant(int k) { g_mutant = k; }
#define MUT(k) (g_mutant == (k))
#else
#define MUT(k) 0
#endif

static char g_err[512] = "";
const char* tgo_last_error(void) { return g_err; }
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

struct tgo_state {
    int mode, n, np, ng, nr, ncons, has_cmm;
    int C, S, use_drude_chains, use_com;
    double kB, T, tau, TD, tauD, dt, max_dist;
    double realkbT, drudekbT;
    double *mass, *inv_mass;
    int *pd, *pp;              /* pair drude / parent */
    int *group, *resid;
    int *normal, nnormal;
    double *pair_inv_total, *pair_inv_reduced;   /* Ref :131-132 */
    /* TGNH residue table: (count, first) K :90-91, Cu :121-125 */
    int *res_count, *res_first;
    double *res_inv_mass;      /* API :147-153 */
    /* --- dualNH thermostat (Ref) --- */
    double realDof, drudeDof, realNkbT, drudeNkbT;
    int numTempGroup, idxMaxNHChains, iNumNHChains;
    /* --- TGNH thermostat (Cu) --- */
    double *tgDof;             /* [G+2] */
    double *tgRed;             /* [G+1] */
    double *tgNkbT;            /* [G+2] */
    /* chain arrays; layout depends on mode */
    double *eta, *etaDot, *etaDotDot, *etaMass;
    int len_eta, len_etaDot, len_etaDotDot, len_etaMass;
    double KESum;
    double time;
    long step_count;
    /* scratch */
    double *comv;   /* [R][4] (vx,vy,vz,w=1/M) */
    double *normv;  /* [N][3] */
    /* harness call-outs */
    int ncl; int *cl_atoms, *cl_ncons, *cl_pairs; double *cl_dist;
    int nvs; int *vs_atoms; double *vs_w;
};

static void* xcalloc(size_t n, size_t sz) {
    void* p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "tgnh_oracle: out of memory\n"); abort(); }
    return p;
}

static int cmp_int(const void* a, const void* b) {
    int x = *(const int*)a, y = *(const int*)b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------ */
/* init, dualNH: Ref :104-219                                          */
/* ------------------------------------------------------------------ */
static void init_dualnh(tgo_state* s) {
    int C = s->C;
    s->realDof = 0; s->drudeDof = 0;
    for (int i = 0; i < s->n; i++)                      /* Ref :114-120 */
        s->realDof += (s->mass[i] == 0.0 ? 0 : 3);
    for (int i = 0; i < s->np; i++) {                   /* Ref :121-135 */
        s->realDof -= 3;
        s->drudeDof += 3;
    }
    if (s->use_drude_chains) {                          /* Ref :139-154 */
        s->numTempGroup = 2;
        s->idxMaxNHChains = C * 2 - 1;
        s->iNumNHChains = C * 2;
    } else {
        s->numTempGroup = 1;
        s->idxMaxNHChains = C * 1;
        s->iNumNHChains = C * 1 + 1;
    }
    s->realDof -= s->ncons;                             /* Ref :157 */
    if (s->has_cmm && !MUT(1)) s->realDof -= 3;         /* Ref :158-165 */
    if (MUT(2)) s->realDof -= 3;
    s->realNkbT = s->realDof * s->realkbT;              /* Ref :168-171 */
    s->drudeNkbT = s->drudeDof * s->drudekbT;

    int ntg = s->numTempGroup;
    int n_eta = s->use_drude_chains ? 2 * C : C + 1;
    s->len_eta = n_eta; s->len_etaDotDot = n_eta; s->len_etaMass = n_eta;
    s->len_etaDot = n_eta + 2;                          /* Ref :216-217 dummies */
    s->eta = xcalloc(n_eta, sizeof(double));
    s->etaDot = xcalloc(n_eta + 2, sizeof(double));
    s->etaDotDot = xcalloc(n_eta, sizeof(double));
    s->etaMass = xcalloc(n_eta, sizeof(double));
    s->etaMass[0] = s->realNkbT * pow(s->tau, 2);
    s->etaMass[1] = s->drudeNkbT * pow(s->tauD, 2);
    if (s->use_drude_chains) {                          /* Ref :192-205 */
        for (int ich = 1; ich < C; ich++) {
            s->etaMass[2 * ich] = s->realkbT * pow(s->tau, 2);
            s->etaMass[2 * ich + 1] = s->drudekbT * pow(s->tauD, 2);
            s->etaDotDot[ich * ntg] = (s->etaMass[(ich - 1) * ntg] * s->etaDot[(ich - 1) * ntg] * s->etaDot[(ich - 1) * ntg] - s->realkbT) / s->etaMass[ich * ntg];
            s->etaDotDot[ich * ntg + 1] = (s->etaMass[(ich - 1) * ntg + 1] * s->etaDot[(ich - 1) * ntg + 1] * s->etaDot[(ich - 1) * ntg + 1] - s->drudekbT) / s->etaMass[ich * ntg + 1];
        }
    } else {                                            /* Ref :206-214 */
        for (int ich = 1; ich < C; ich++) {
            s->etaMass[ich + 1] = s->realkbT * pow(s->tau, 2);
            s->etaDotDot[ich * ntg + 1] = (s->etaMass[(ich - 1) * ntg + 1] * s->etaDot[(ich - 1) * ntg + 1] * s->etaDot[(ich - 1) * ntg + 1] - s->realkbT) / s->etaMass[ich * ntg + 1];
        }
    }
}

/* ------------------------------------------------------------------ */
/* init, TGNH: Cu :75-235 (+ API :136-153 for the residue table)       */
/* ------------------------------------------------------------------ */
static int init_tgnh(tgo_state* s, const tgo_desc* d) {
    int G = s->ng, C = s->C, R = s->nr;
    s->res_count = xcalloc(R, sizeof(int));
    s->res_first = xcalloc(R, sizeof(int));
    s->res_inv_mass = xcalloc(R, sizeof(double));
    for (int r = 0; r < R; r++) s->res_first[r] = -1;   /* Cu :88-89 */
    {   /* API :147-153 residueMasses over ALL particles */
        double* rm = xcalloc(R, sizeof(double));
        for (int i = 0; i < s->n; i++) rm[s->resid[i]] += s->mass[i];
        for (int r = 0; r < R; r++) s->res_inv_mass[r] = 1.0 / rm[r];
        free(rm);
    }
    s->tgDof = xcalloc(G + 2, sizeof(double));
    s->tgRed = xcalloc(G + 1, sizeof(double));
    s->tgNkbT = xcalloc(G + 2, sizeof(double));
    int prevRes = -1;
    for (int i = 0; i < s->n; i++) {                    /* Cu :114-134 */
        int tg = s->group[i], resid = s->resid[i];
        s->res_count[resid] += 1;
        if (prevRes != resid) { s->res_first[resid] = i; prevRes = resid; }
        if (s->mass[i] != 0.0) {
            s->tgDof[tg] += 3;
            if (s->use_com && !MUT(3)) s->tgRed[tg] += 3 * s->mass[i] * s->res_inv_mass[resid];
        }
    }
    double drudeDof = 0;
    for (int i = 0; i < s->np; i++) {                   /* Cu :135-150 */
        int tg = s->group[s->pd[i]], tg1 = s->group[s->pp[i]];
        if (tg != tg1)
            return fail(TGO_ERR_GROUP_MISMATCH, "Temperature group for drude particle must be the same as the parent particle");
        s->tgDof[tg] -= 3;
        drudeDof += 3;
    }
    for (int i = 0; i < s->ncons; i++) {                /* Cu :186-196 */
        if (d->constraint_i && d->constraint_j) {
            int tg = s->group[d->constraint_i[i]], tg1 = s->group[d->constraint_j[i]];
            if (tg != tg1)
                return fail(TGO_ERR_GROUP_MISMATCH, "Temperature group of constrained particles must be the same");
            s->tgDof[tg] -= 1;
        } else {
            s->tgDof[0] -= 1;
        }
    }
    if (s->use_com) s->tgDof[G] = 3 * R;                /* Cu :197-199 */
    s->tgDof[G + 1] = drudeDof;                         /* Cu :201 */
    if (s->use_com && s->has_cmm && !MUT(1)) s->tgDof[G] -= 3;     /* Cu :204-212 */
    if (MUT(2)) s->tgDof[0] -= 3;
    s->drudeDof = drudeDof;
    s->drudeNkbT = drudeDof * s->drudekbT;              /* Cu :215 */
    double drudeUnit = s->drudekbT * pow(s->tauD, 2);   /* Cu :216-217 */
    double realUnit = s->realkbT * pow(s->tau, 2);

    s->len_eta = (G + 2) * C; s->len_etaDotDot = (G + 2) * C; s->len_etaMass = (G + 2) * C;
    s->len_etaDot = (G + 2) * (C + 1);                  /* Cu :94-97 */
    s->eta = xcalloc(s->len_eta, sizeof(double));
    s->etaDot = xcalloc(s->len_etaDot, sizeof(double));
    s->etaDotDot = xcalloc(s->len_etaDotDot, sizeof(double));
    s->etaMass = xcalloc(s->len_etaMass, sizeof(double));
#define EM(g, i) s->etaMass[(g) * C + (i)]
#define ED(g, i) s->etaDot[(g) * (C + 1) + (i)]
#define EDD(g, i) s->etaDotDot[(g) * C + (i)]
#define ETA(g, i) s->eta[(g) * C + (i)]
    for (int i = 0; i < G + 1; i++) {                   /* Cu :218-225 */
        s->tgNkbT[i] = (s->tgDof[i] - s->tgRed[i]) * s->realkbT;
        EM(i, 0) = (s->tgDof[i] - s->tgRed[i]) * realUnit;
        for (int ich = 1; ich < C; ich++) {
            EM(i, ich) = realUnit;
            EDD(i, ich) = (EM(i, ich - 1) * ED(i, ich - 1) * ED(i, ich - 1) - s->realkbT) / EM(i, ich);
        }
    }
    int itg = G + 1;                                    /* Cu :227-235 */
    s->tgNkbT[itg] = s->drudeNkbT;
    EM(itg, 0) = drudeDof * drudeUnit;
    for (int ich = 1; ich < C; ich++) {
        EM(itg, ich) = drudeUnit;
        if (s->use_drude_chains)
            EDD(itg, ich) = (EM(itg, ich - 1) * ED(itg, ich - 1) * ED(itg, ich - 1) - s->drudekbT) / EM(itg, ich);
    }
    s->comv = xcalloc((size_t)R * 4, sizeof(double));
    s->normv = xcalloc((size_t)s->n * 3, sizeof(double));
    return TGO_OK;
}

int tgo_create(const tgo_desc* d, tgo_state** out) {
    if (!d || !out) return fail(TGO_ERR_ARG, "null argument");
    if (d->num_particles < 0 || d->num_pairs < 0 || d->num_nh_chains < 1 || d->drude_steps_per_real_step < 1)
        return fail(TGO_ERR_ARG, "bad sizes");
    if (d->mode == TGO_MODE_TGNH && (d->num_groups < 1 || d->num_residues < 1 || !d->group || !d->resid))
        return fail(TGO_ERR_ARG, "TGNH mode needs groups and residues");
    tgo_state* s = xcalloc(1, sizeof *s);
    s->mode = d->mode; s->n = d->num_particles; s->np = d->num_pairs;
    s->ng = d->num_groups; s->nr = d->num_residues; s->ncons = d->num_constraints;
    s->has_cmm = d->has_cm_motion_remover;
    s->C = d->num_nh_chains; s->S = d->drude_steps_per_real_step;
    s->use_drude_chains = d->use_drude_nh_chains; s->use_com = d->use_com_temp_group;
    s->kB = d->kB; s->T = d->temperature; s->tau = d->coupling_time;
    s->TD = d->drude_temperature; s->tauD = d->drude_coupling_time;
    s->dt = d->step_size; s->max_dist = d->max_drude_distance;
    s->realkbT = s->kB * s->T;                          /* Ref :107-108, Cu :80-81 */
    s->drudekbT = s->kB * s->TD;
    int N = s->n, P = s->np;
    s->mass = xcalloc(N, sizeof(double));
    s->inv_mass = xcalloc(N, sizeof(double));
    for (int i = 0; i < N; i++) {
        s->mass[i] = d->mass[i];
        s->inv_mass[i] = (d->mass[i] == 0.0 ? 0.0 : 1.0 / d->mass[i]);   /* Ref :118 */
    }
    s->pd = xcalloc(P, sizeof(int)); s->pp = xcalloc(P, sizeof(int));
    s->pair_inv_total = xcalloc(P, sizeof(double));
    s->pair_inv_reduced = xcalloc(P, sizeof(double));
    char* in_pair = xcalloc(N, 1);
    for (int i = 0; i < P; i++) {
        int p = d->pair_drude[i], p1 = d->pair_parent[i];
        if (p < 0 || p >= N || p1 < 0 || p1 >= N) { free(in_pair); tgo_destroy(s); return fail(TGO_ERR_ARG, "pair index out of range"); }
        s->pd[i] = p; s->pp[i] = p1;
        in_pair[p] = 1; in_pair[p1] = 1;                /* Ref :125-126 set erase */
        double m1 = s->mass[p], m2 = s->mass[p1];
        s->pair_inv_total[i] = 1.0 / (m1 + m2);         /* Ref :131 */
        s->pair_inv_reduced[i] = (m1 + m2) / (m1 * m2); /* Ref :132 */
    }
    s->normal = xcalloc(N, sizeof(int));
    s->nnormal = 0;
    for (int i = 0; i < N; i++)                         /* Ref :137 (ascending: std::set) */
        if (!in_pair[i]) s->normal[s->nnormal++] = i;
    free(in_pair);
    qsort(s->normal, s->nnormal, sizeof(int), cmp_int); /* already sorted; explicit */
    if (d->group) { s->group = xcalloc(N, sizeof(int)); memcpy(s->group, d->group, N * sizeof(int)); }
    if (d->resid) { s->resid = xcalloc(N, sizeof(int)); memcpy(s->resid, d->resid, N * sizeof(int)); }
    int rc = TGO_OK;
    if (s->mode == TGO_MODE_DUALNH) {
        init_dualnh(s);
    } else {
        for (int i = 0; i < N; i++) {
            if (s->group[i] < 0 || s->group[i] >= s->ng || s->resid[i] < 0 || s->resid[i] >= s->nr) {
                tgo_destroy(s); return fail(TGO_ERR_ARG, "group/residue index out of range");
            }
        }
        rc = init_tgnh(s, d);
    }
    if (rc != TGO_OK) { tgo_destroy(s); return rc; }
    *out = s;
    return TGO_OK;
}

void tgo_destroy(tgo_state* s) {
    if (!s) return;
    free(s->mass); free(s->inv_mass); free(s->pd); free(s->pp); free(s->group); free(s->resid);
    free(s->normal); free(s->pair_inv_total); free(s->pair_inv_reduced);
    free(s->res_count); free(s->res_first); free(s->res_inv_mass);
    free(s->tgDof); free(s->tgRed); free(s->tgNkbT);
    free(s->eta); free(s->etaDot); free(s->etaDotDot); free(s->etaMass);
    free(s->comv); free(s->normv);
    free(s->cl_atoms); free(s->cl_ncons); free(s->cl_pairs); free(s->cl_dist); free(s->vs_atoms); free(s->vs_w);
    free(s);
}

void tgo_set_step_size(tgo_state* s, double dt) { s->dt = dt; }
void tgo_set_drude_steps(tgo_state* s, int n) { s->S = n; }
void tgo_set_max_drude_distance(tgo_state* s, double d) { s->max_dist = d; }
int tgo_num_normal(const tgo_state* s) { return s->nnormal; }
void tgo_get_normal(const tgo_state* s, int* out) { memcpy(out, s->normal, s->nnormal * sizeof(int)); }
int tgo_num_thermostats(const tgo_state* s) { return s->mode == TGO_MODE_DUALNH ? 2 : s->ng + 2; }

void tgo_get_dof(const tgo_state* s, double* dof, double* nkt) {
    if (s->mode == TGO_MODE_DUALNH) {
        dof[0] = s->realDof; dof[1] = s->drudeDof;
        nkt[0] = s->realNkbT; nkt[1] = s->drudeNkbT;
    } else {
        for (int i = 0; i < s->ng + 2; i++) {          /* Cu :219, :241 */
            dof[i] = s->tgDof[i] - (i <= s->ng ? s->tgRed[i] : 0.0);
            nkt[i] = s->tgNkbT[i];
        }
    }
}

static double* chain_arr(const tgo_state* s, int which, int* len) {
    switch (which) {
        case 0: *len = s->len_eta; return s->eta;
        case 1: *len = s->len_etaDot; return s->etaDot;
        case 2: *len = s->len_etaDotDot; return s->etaDotDot;
        default: *len = s->len_etaMass; return s->etaMass;
    }
}
int tgo_chain_len(const tgo_state* s, int which) { int l; chain_arr(s, which, &l); return l; }
void tgo_get_chain(const tgo_state* s, int which, double* out) { int l; double* a = chain_arr(s, which, &l); memcpy(out, a, l * sizeof(double)); }
void tgo_set_chain(tgo_state* s, int which, const double* in) { int l; double* a = chain_arr(s, which, &l); memcpy(a, in, l * sizeof(double)); }

#define V(a, i, j) (a)[3 * (size_t)(i) + (j)]
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* ------------------------------------------------------------------ */
/* A3: KE, dualNH.  Ref :439-460                                       */
/* ------------------------------------------------------------------ */
static void ke_dualnh(const tgo_state* s, const double* vel, double* ke) {
    double realKE = 0.0, drudeKE = 0.0;
    for (int i = 0; i < s->nnormal; i++) {              /* Ref :443-448 */
        int index = s->normal[i];
        if (s->inv_mass[index] != 0)
            realKE += dot3(&V(vel, index, 0), &V(vel, index, 0)) / s->inv_mass[index];
    }
    for (int i = 0; i < s->np; i++) {                   /* Ref :451-460 */
        int p1 = s->pd[i], p2 = s->pp[i];
        double m1f = s->pair_inv_total[i] / s->inv_mass[p1];
        double m2f = s->pair_inv_total[i] / s->inv_mass[p2];
        double cm[3], rel[3];
        for (int j = 0; j < 3; j++) {
            cm[j] = V(vel, p1, j) * m1f + V(vel, p2, j) * m2f;
            rel[j] = V(vel, p2, j) - V(vel, p1, j);
        }
        realKE += dot3(cm, cm) / s->pair_inv_total[i];
        drudeKE += dot3(rel, rel) / s->pair_inv_reduced[i];
    }
    ke[0] = realKE; ke[1] = drudeKE;
}

/* ------------------------------------------------------------------ */
/* A4: COM velocities + normalized velocities + KE, TGNH.              */
/* K :82-113 (calcCOMVelocities), :119-133 (normalizeVelocities),      */
/* :138-200 (computeNormalizedKineticEnergies)                         */
/* ------------------------------------------------------------------ */
static void com_and_norm(const tgo_state* s, const double* vel, double* comv, double* normv) {
    for (int r = 0; r < s->nr; r++) {                   /* K :86-111 */
        double* c = comv + 4 * (size_t)r;
        c[0] = c[1] = c[2] = c[3] = 0;
        if (s->use_com) {
            double comMass = 0.0;
            for (int j = 0; j < s->res_count[r]; j++) {
                int index = s->res_first[r] + j;
                double w = s->inv_mass[index];
                if (w != 0) {
                    double m = 1.0 / w;
                    c[0] += V(vel, index, 0) * m;
                    c[1] += V(vel, index, 1) * m;
                    c[2] += V(vel, index, 2) * m;
                    comMass += m;
                }
            }
            c[3] = 1.0 / comMass;
            c[0] *= c[3]; c[1] *= c[3]; c[2] *= c[3];
        } else {
            c[3] = 1.0;
        }
    }
    for (int i = 0; i < s->n; i++) {                    /* K :123-129 */
        const double* c = comv + 4 * (size_t)s->resid[i];
        for (int j = 0; j < 3; j++) V(normv, i, j) = V(vel, i, j) - (MUT(8) ? 0.0 : c[j]);
    }
}

static void ke_tgnh_from_norm(const tgo_state* s, const double* comv, const double* normv, double* ke) {
    int G = s->ng;
    for (int i = 0; i < G + 2; i++) ke[i] = 0;
    for (int r = 0; r < s->nr; r++) {                   /* K :152-158 */
        const double* c = comv + 4 * (size_t)r;
        ke[G] += (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / c[3];
    }
    for (int i = 0; i < s->nnormal; i++) {              /* K :161-168 */
        int index = s->normal[i];
        double w = s->inv_mass[index];
        if (w != 0)
            ke[s->group[index]] += dot3(&V(normv, index, 0), &V(normv, index, 0)) / w;
    }
    for (int i = 0; i < s->np; i++) {                   /* K :171-186 */
        int px = s->pd[i], py = s->pp[i];
        double w1 = s->inv_mass[px], w2 = s->inv_mass[py];
        double mass1 = 1.0 / w1, mass2 = 1.0 / w2;
        double invTotalMass = 1.0 / (mass1 + mass2);
        double invReducedMass = (mass1 + mass2) * w1 * w2;
        double m1f = invTotalMass * mass1, m2f = invTotalMass * mass2;
        double cm[3], rel[3];
        for (int j = 0; j < 3; j++) {
            cm[j] = V(normv, px, j) * m1f + V(normv, py, j) * m2f;
            rel[j] = V(normv, py, j) - V(normv, px, j);
        }
        ke[s->group[px]] += dot3(cm, cm) * (mass1 + mass2);
        ke[G + 1] += dot3(rel, rel) * (1.0 / invReducedMass);
    }
}

void tgo_kinetic_energies(const tgo_state* s, const double* vel, double* ke) {
    if (s->mode == TGO_MODE_DUALNH) { ke_dualnh(s, vel, ke); return; }
    com_and_norm(s, vel, s->comv, s->normv);
    ke_tgnh_from_norm(s, s->comv, s->normv, ke);
}

/* ------------------------------------------------------------------ */
/* A5: chain, dualNH.  Ref :467-504 (bug-compatible indexing)          */
/* ------------------------------------------------------------------ */
static void chain_dualnh(tgo_state* s, const double* ke_in, double* scale) {
    const double dt = s->dt;
    const double dtc = dt / s->S;                       /* Ref :432-435 */
    const double dtc2 = dtc / 2.0, dtc4 = dtc / 4.0, dtc8 = dtc / 8.0;
    double realKE = ke_in[0], drudeKE = ke_in[1];
    double scaleReal = 1.0, scaleDrude = 1.0, expfac = 1.0;
    double *etaDot = s->etaDot, *etaDotDot = s->etaDotDot, *eta = s->eta, *etaMass = s->etaMass;
    etaDotDot[0] = (realKE - s->realNkbT) / etaMass[0]; /* Ref :471-472 */
    etaDotDot[1] = (drudeKE - s->drudeNkbT) / etaMass[1];
    for (int iter = 0; iter < s->S; iter++) {           /* Ref :474 */
        for (int i = s->idxMaxNHChains; i >= 0; i--) {  /* Ref :476-481 */
            expfac = exp(-dtc8 * etaDot[i + (MUT(5) ? 2 : s->numTempGroup)]);
            etaDot[i] *= expfac;
            etaDot[i] += etaDotDot[i] * dtc4;
            etaDot[i] *= expfac;
        }
        scaleReal *= exp(-(MUT(9) ? dtc : dtc2) * etaDot[0]);   /* Ref :483-486 */
        scaleDrude *= exp(-(MUT(9) ? dtc : dtc2) * etaDot[1]);
        realKE *= exp(-dtc * etaDot[0]);
        drudeKE *= exp(-dtc * etaDot[1]);
        for (int i = 0; i < s->iNumNHChains; i++)       /* Ref :487-489 */
            eta[i] += dtc2 * etaDot[i];
        etaDotDot[0] = (realKE - s->realNkbT) / etaMass[0];     /* Ref :491-492 */
        etaDotDot[1] = (drudeKE - s->drudeNkbT) / etaMass[1];
        for (int i = 0; i < s->iNumNHChains; i++) {     /* Ref :494-503 */
            expfac = exp(-dtc8 * etaDot[i + (MUT(4) ? s->numTempGroup : 2)]);
            etaDot[i] *= expfac;
            if (i > 1) {
                double dofkbT = ((i % 2 == 0) != MUT(6) ? s->realkbT : s->drudekbT);
                etaDotDot[i] = (etaMass[i - 2] * etaDot[i - 2] * etaDot[i - 2] - dofkbT) / etaMass[i];
            }
            etaDot[i] += etaDotDot[i] * dtc4;
            etaDot[i] *= expfac;
        }
    }
    scale[0] = scaleReal; scale[1] = scaleDrude;
}

/* ------------------------------------------------------------------ */
/* A5: chain, TGNH.  Cu :558-650                                       */
/* ------------------------------------------------------------------ */
static void chain_tgnh(tgo_state* s, const double* ke_in, double* scale) {
    int G = s->ng, C = s->C;
    const double dtc = s->dt / s->S;                    /* Cu :440-443 */
    const double dtc2 = dtc / 2.0, dtc4 = dtc / 4.0, dtc8 = dtc / 8.0;
    double* ke = xcalloc(G + 2, sizeof(double));
    for (int i = 0; i < G + 2; i++) { ke[i] = ke_in[i]; scale[i] = 1.0; }
    s->KESum = 0.0;                                     /* Cu :493-497 */
    for (int i = 0; i < G + 2; i++) s->KESum += ke[i];
    s->KESum = 0.5 * s->KESum;
    for (int itg = 0; itg < G + 1; itg++) {             /* Cu :560-595 */
        double expfac = 1.0;
        if (EM(itg, 0) > 0)
            EDD(itg, 0) = (ke[itg] - s->tgNkbT[itg]) / EM(itg, 0);
        for (int iter = 0; iter < s->S; iter++) {
            for (int i = C - 1; i >= 0; i--) {          /* Cu :566-571 */
                expfac = exp(-dtc8 * ED(itg, i + 1));
                ED(itg, i) *= expfac;
                ED(itg, i) += EDD(itg, i) * dtc4;
                ED(itg, i) *= expfac;
            }
            scale[itg] *= exp(-(MUT(9) ? dtc : dtc2) * ED(itg, 0));      /* Cu :573-574 */
            ke[itg] *= exp(-dtc * ED(itg, 0));
            for (int i = 0; i < C; i++)                 /* Cu :575-577 */
                ETA(itg, i) += dtc2 * ED(itg, i);
            if (EM(itg, 0) > 0)                         /* Cu :579-581 */
                EDD(itg, 0) = (ke[itg] - s->tgNkbT[itg]) / EM(itg, 0);
            if (MUT(7)) expfac = 1.0;
            ED(itg, 0) *= expfac;                       /* Cu :583-585 (expfac reused) */
            ED(itg, 0) += EDD(itg, 0) * dtc4;
            ED(itg, 0) *= expfac;
            for (int i = 1; i < C; i++) {               /* Cu :586-592 */
                expfac = exp(-dtc8 * ED(itg, i + 1));
                ED(itg, i) *= expfac;
                EDD(itg, i) = (EM(itg, i - 1) * ED(itg, i - 1) * ED(itg, i - 1) - (MUT(6) ? s->drudekbT : s->realkbT)) / EM(itg, i);
                ED(itg, i) += EDD(itg, i) * dtc4;
                ED(itg, i) *= expfac;
            }
        }
    }
    int itg = G + 1;                                    /* Cu :597-642 */
    double expfac = 1.0;
    EDD(itg, 0) = (ke[itg] - s->tgNkbT[itg]) / EM(itg, 0);
    for (int iter = 0; iter < s->S; iter++) {
        if (s->use_drude_chains) {
            for (int i = C - 1; i > 0; i--) {
                expfac = exp(-dtc8 * ED(itg, i + 1));
                ED(itg, i) *= expfac;
                ED(itg, i) += EDD(itg, i) * dtc4;
                ED(itg, i) *= expfac;
            }
        }
        expfac = exp(-dtc8 * ED(itg, 1));
        ED(itg, 0) *= expfac;
        ED(itg, 0) += EDD(itg, 0) * dtc4;
        ED(itg, 0) *= expfac;
        scale[itg] *= exp(-(MUT(9) ? dtc : dtc2) * ED(itg, 0));
        ke[itg] *= exp(-dtc * ED(itg, 0));
        ETA(itg, 0) += dtc2 * ED(itg, 0);
        if (s->use_drude_chains)
            for (int i = 1; i < C; i++) ETA(itg, i) += dtc2 * ED(itg, i);
        EDD(itg, 0) = (ke[itg] - s->tgNkbT[itg]) / EM(itg, 0);
        ED(itg, 0) *= expfac;
        ED(itg, 0) += EDD(itg, 0) * dtc4;
        ED(itg, 0) *= expfac;
        if (s->use_drude_chains) {
            for (int i = 1; i < C; i++) {
                expfac = exp(-dtc8 * ED(itg, i + 1));
                ED(itg, i) *= expfac;
                EDD(itg, i) = (EM(itg, i - 1) * ED(itg, i - 1) * ED(itg, i - 1) - s->drudekbT) / EM(itg, i);
                ED(itg, i) += EDD(itg, i) * dtc4;
                ED(itg, i) *= expfac;
            }
        }
    }
    free(ke);
}

void tgo_chain_only(tgo_state* s, const double* ke_in, double* scale_out) {
    if (s->mode == TGO_MODE_DUALNH) chain_dualnh(s, ke_in, scale_out);
    else chain_tgnh(s, ke_in, scale_out);
}

/* ------------------------------------------------------------------ */
/* A6: rescale.  dualNH Ref :516-541; TGNH K :249-301                  */
/* ------------------------------------------------------------------ */
static void scale_dualnh(const tgo_state* s, double* vel, const double* scale) {
    double scaleReal = scale[0], scaleDrude = scale[1];
    for (int i = 0; i < s->nnormal; i++) {              /* Ref :517-524 */
        int index = s->normal[i];
        if (s->inv_mass[index] != 0.0)
            for (int j = 0; j < 3; j++) V(vel, index, j) = scaleReal * V(vel, index, j);
    }
    for (int i = 0; i < s->np; i++) {                   /* Ref :527-541 */
        int p1 = s->pd[i], p2 = s->pp[i];
        double m1f = s->pair_inv_total[i] / s->inv_mass[p1];
        double m2f = s->pair_inv_total[i] / s->inv_mass[p2];
        double cm[3], rel[3];
        for (int j = 0; j < 3; j++) {
            cm[j] = V(vel, p1, j) * m1f + V(vel, p2, j) * m2f;
            rel[j] = V(vel, p2, j) - V(vel, p1, j);
        }
        double scaleCM = scaleReal;
        for (int j = 0; j < 3; j++) { cm[j] = scaleCM * cm[j]; rel[j] = scaleDrude * rel[j]; }
        for (int j = 0; j < 3; j++) {
            V(vel, p1, j) = cm[j] - rel[j] * m2f;
            V(vel, p2, j) = cm[j] + rel[j] * m1f;
        }
    }
}

static void scale_tgnh(const tgo_state* s, double* vel, const double* normv, const double* scale) {
    int G = s->ng;
    double vscaleCOM = scale[G], vscaleDrude = scale[G + 1];    /* K :252-253 */
    for (int i = 0; i < s->nnormal; i++) {              /* K :255-266 */
        int index = s->normal[i];
        double vscale = scale[s->group[index]];
        if (s->inv_mass[index] != 0) {
            for (int j = 0; j < 3; j++) {
                double vr = V(normv, index, j);
                V(vel, index, j) = vscale * vr + vscaleCOM * (V(vel, index, j) - vr);
            }
        }
    }
    for (int i = 0; i < s->np; i++) {                   /* K :270-300 */
        int px = s->pd[i], py = s->pp[i];
        double vscaleCM = scale[s->group[px]];
        double w1 = s->inv_mass[px], w2 = s->inv_mass[py];
        double mass1 = 1.0 / w1, mass2 = 1.0 / w2;
        double invTotalMass = 1.0 / (mass1 + mass2);
        double m1f = invTotalMass * mass1, m2f = invTotalMass * mass2;
        for (int j = 0; j < 3; j++) {
            double r1 = V(normv, px, j), r2 = V(normv, py, j);
            double c1 = V(vel, px, j) - r1, c2 = V(vel, py, j) - r2;
            double cm = r1 * m1f + r2 * m2f;
            double rel = r2 - r1;
            cm = vscaleCM * cm;
            rel = vscaleDrude * rel;
            V(vel, px, j) = cm - rel * m2f + vscaleCOM * c1;
            V(vel, py, j) = cm + rel * m1f + vscaleCOM * c2;
        }
    }
}

void tgo_scale_velocities(const tgo_state* s, double* vel, const double* scale) {
    if (s->mode == TGO_MODE_DUALNH) { scale_dualnh(s, vel, scale); return; }
    com_and_norm(s, vel, s->comv, s->normv);
    scale_tgnh(s, vel, s->normv, scale);
}

/* A3..A6.  Ref :426-546 ; Cu :433-652 + :351-353 */
void tgo_propagate_nhc(tgo_state* s, double* vel, double* ke_out, double* scale_out) {
    int n = tgo_num_thermostats(s);
    double* ke = xcalloc(n, sizeof(double));
    double* sc = xcalloc(n, sizeof(double));
    if (s->mode == TGO_MODE_DUALNH) {
        ke_dualnh(s, vel, ke);
        chain_dualnh(s, ke, sc);
        scale_dualnh(s, vel, sc);
    } else {
        com_and_norm(s, vel, s->comv, s->normv);
        ke_tgnh_from_norm(s, s->comv, s->normv, ke);
        chain_tgnh(s, ke, sc);
        scale_tgnh(s, vel, s->normv, sc);
    }
    if (ke_out) memcpy(ke_out, ke, n * sizeof(double));
    if (scale_out) memcpy(scale_out, sc, n * sizeof(double));
    free(ke); free(sc);
}

/* ------------------------------------------------------------------ */
/* A7: half kick.  Ref :548-584 ; K :307-365 (fscale = dt/2, forces    */
/* here are plain doubles, not 2^32 fixed point)                        */
/* ------------------------------------------------------------------ */
void tgo_half_kick(const tgo_state* s, double* vel, const double* force) {
    double dt = s->dt;
    for (int i = 0; i < s->nnormal; i++) {              /* Ref :555-562 */
        int index = s->normal[i];
        double invMass = s->inv_mass[index];
        if (invMass != 0.0)
            for (int j = 0; j < 3; j++) V(vel, index, j) += 0.5 * dt * invMass * V(force, index, j);
    }
    for (int i = 0; i < s->np; i++) {                   /* Ref :565-583 */
        int p1 = s->pd[i], p2 = s->pp[i];
        double invTot, invRed, m1f, m2f;
        if (s->mode == TGO_MODE_DUALNH) {
            invTot = s->pair_inv_total[i]; invRed = s->pair_inv_reduced[i];
            m1f = invTot / s->inv_mass[p1]; m2f = invTot / s->inv_mass[p2];
        } else {                                        /* K :334-339 */
            double w1 = s->inv_mass[p1], w2 = s->inv_mass[p2];
            double mass1 = 1.0 / w1, mass2 = 1.0 / w2;
            invTot = 1.0 / (mass1 + mass2);
            invRed = (mass1 + mass2) * w1 * w2;
            m1f = invTot * mass1; m2f = invTot * mass2;
        }
        for (int j = 0; j < 3; j++) {
            double cm = V(vel, p1, j) * m1f + V(vel, p2, j) * m2f;
            double rel = V(vel, p2, j) - V(vel, p1, j);
            double cmF = V(force, p1, j) + V(force, p2, j);
            double relF = V(force, p2, j) * m1f - V(force, p1, j) * m2f;
            cm += 0.5 * dt * invTot * cmF;
            rel += 0.5 * dt * invRed * relF;
            V(vel, p1, j) = cm - rel * m2f;
            V(vel, p2, j) = cm + rel * m1f;
        }
    }
}

/* ------------------------------------------------------------------ */
/* A8: drift.  Ref :253-258, :278-284 ; K :322-324,:360-363,:435-466   */
/* ------------------------------------------------------------------ */
void tgo_drift(const tgo_state* s, double* pos, double* vel) {
    double dt = s->dt;
    if (s->mode == TGO_MODE_DUALNH) {
        double dtInv = 1.0 / dt;
        for (int i = 0; i < s->n; i++) {
            if (s->inv_mass[i] != 0.0) {
                for (int j = 0; j < 3; j++) {
                    double xp = V(pos, i, j) + V(vel, i, j) * dt;      /* Ref :258 */
                    V(vel, i, j) = (xp - V(pos, i, j)) * dtInv;        /* Ref :281 */
                    V(pos, i, j) = xp;                                 /* Ref :282 */
                }
            }
        }
    } else {
        double invStepSize = 1.0 / dt;                  /* K :436 */
        for (int i = 0; i < s->n; i++) {
            if (s->inv_mass[i] != 0) {                  /* K :440 */
                for (int j = 0; j < 3; j++) {
                    double delta = dt * V(vel, i, j);   /* K :323, :361-362 */
                    V(pos, i, j) += delta;              /* K :450-452 */
                    V(vel, i, j) = invStepSize * delta; /* K :453-455 */
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* A10: hard wall.  Ref :298-363 ; K :471-574                          */
/* ------------------------------------------------------------------ */
int tgo_hardwall(const tgo_state* s, double* pos, double* vel) {
    const double maxDrudeDistance = s->max_dist;
    if (!(maxDrudeDistance > 0)) return TGO_OK;         /* Ref :299 ; Cu :372 */
    const double dt = s->dt;
    const double hardwallscaleDrude = sqrt(s->kB * s->TD);   /* Ref :300 ; Cu :299 */
    for (int i = 0; i < s->np; i++) {
        int p1 = s->pd[i], p2 = s->pp[i];
        double delta[3];
        for (int j = 0; j < 3; j++) delta[j] = V(pos, p1, j) - V(pos, p2, j);
        double r = sqrt(dot3(delta, delta));
        double rInv = 1 / r;
        if (rInv * maxDrudeDistance < 1.0) {
            if (s->mode == TGO_MODE_DUALNH && rInv * maxDrudeDistance < 0.5)   /* Ref :311-312 (K has no throw) */
                return fail(TGO_ERR_HARDWALL, "Drude particle moved too far beyond hard wall constraint");
            double bondDir[3], vb1[3], vp1[3];
            for (int j = 0; j < 3; j++) bondDir[j] = delta[j] * rInv;
            double mass1 = s->mass[p1], mass2 = s->mass[p2];
            double deltaR = r - maxDrudeDistance;
            double deltaT = dt;
            double dotvr1 = dot3(&V(vel, p1, 0), bondDir);
            for (int j = 0; j < 3; j++) { vb1[j] = bondDir[j] * dotvr1; vp1[j] = V(vel, p1, j) - vb1[j]; }
            if (mass2 == 0) {                           /* Ref :323-334 */
                if (dotvr1 != 0.0) deltaT = deltaR / fabs(dotvr1);
                if (deltaT > dt) deltaT = dt;
                dotvr1 = -dotvr1 * hardwallscaleDrude / (fabs(dotvr1) * sqrt(mass1));
                double dr = -deltaR + deltaT * dotvr1;
                for (int j = 0; j < 3; j++) {
                    V(pos, p1, j) += bondDir[j] * dr;
                    V(vel, p1, j) = vp1[j] + bondDir[j] * dotvr1;
                }
            } else {                                    /* Ref :335-360 */
                double invTotalMass = (s->mode == TGO_MODE_DUALNH) ? s->pair_inv_total[i] : 1.0 / (mass1 + mass2);
                double vb2[3], vp2[3];
                double dotvr2 = dot3(&V(vel, p2, 0), bondDir);
                for (int j = 0; j < 3; j++) { vb2[j] = bondDir[j] * dotvr2; vp2[j] = V(vel, p2, j) - vb2[j]; }
                double vbCMass = (mass1 * dotvr1 + mass2 * dotvr2) * invTotalMass;
                dotvr1 -= vbCMass;
                dotvr2 -= vbCMass;
                if (dotvr1 != dotvr2) deltaT = deltaR / fabs(dotvr1 - dotvr2);
                if (deltaT > dt) deltaT = dt;
                double vBond = hardwallscaleDrude / sqrt(mass1);
                dotvr1 = -dotvr1 * vBond * mass2 * invTotalMass / fabs(dotvr1);
                dotvr2 = -dotvr2 * vBond * mass1 * invTotalMass / fabs(dotvr2);
                double dr1 = -deltaR * mass2 * invTotalMass + deltaT * dotvr1;
                double dr2 = deltaR * mass1 * invTotalMass + deltaT * dotvr2;
                dotvr1 += vbCMass;
                dotvr2 += vbCMass;
                for (int j = 0; j < 3; j++) {
                    V(pos, p1, j) += bondDir[j] * dr1;
                    V(pos, p2, j) += bondDir[j] * dr2;
                    V(vel, p1, j) = vp1[j] + bondDir[j] * dotvr1;
                    V(vel, p2, j) = vp2[j] + bondDir[j] * dotvr2;
                }
            }
        }
    }
    return TGO_OK;
}

/* ------------------------------------------------------------------ */
/* A11: orchestration.  Ref :221-415 ; Cu :284-408 (no constraints,    */
/* no virtual sites: those call-outs are no-ops here)                  */
/* ------------------------------------------------------------------ */
int tgo_step_begin(tgo_state* s, double* pos, double* vel, const double* force) {
    tgo_propagate_nhc(s, vel, NULL, NULL);              /* Ref :231 ; Cu :336-353 */
    tgo_half_kick(s, vel, force);                       /* Ref :239 ; Cu :356-360 */
    tgo_drift(s, pos, vel);                             /* Ref :253-284 ; Cu :363-369 */
    return tgo_hardwall(s, pos, vel);                   /* Ref :298-363 ; Cu :372-376 */
}

int tgo_step_end(tgo_state* s, double* vel, const double* force) {
    tgo_half_kick(s, vel, force);                       /* Ref :394 ; Cu :384-388 */
    tgo_propagate_nhc(s, vel, NULL, NULL);              /* Ref :406 ; Cu :394-402 */
    s->time += s->dt;                                   /* Ref :413-414 ; Cu :405-406 */
    s->step_count++;
    return TGO_OK;
}

/* A12.  Cu :654-658 ; Ref :70-98, :586-588 (no constraints) */
double tgo_kinetic_energy_query(const tgo_state* s, const double* vel, const double* force, int ke_sum_valid) {
    if (s->mode == TGO_MODE_TGNH) {
        if (ke_sum_valid) return s->KESum;
        double e = 0.0;
        for (int i = 0; i < s->n; i++)
            if (s->inv_mass[i] != 0) e += dot3(&V(vel, i, 0), &V(vel, i, 0)) / s->inv_mass[i];
        return 0.5 * e;
    }
    double timeShift = 0.5 * s->dt, energy = 0.0;
    for (int i = 0; i < s->n; i++) {
        if (s->inv_mass[i] > 0) {
            double sv[3];
            for (int j = 0; j < 3; j++) sv[j] = V(vel, i, j) + V(force, i, j) * (timeShift * s->inv_mass[i]);
            energy += dot3(sv, sv) / s->inv_mass[i];
        }
    }
    return 0.5 * energy;
}

/* ------------------------------------------------------------------ */
/* Harness force (not from the reference; the test/bench workload)     */
/* ------------------------------------------------------------------ */
void tgo_harness_force(const tgo_state* s, const double* pos, const double* x0,
                       double k_drude, double k_tether, double* force) {
    memset(force, 0, sizeof(double) * 3 * (size_t)s->n);
    for (int i = 0; i < s->nnormal; i++) {
        int index = s->normal[i];
        if (s->mass[index] != 0.0)
            for (int j = 0; j < 3; j++) V(force, index, j) = -k_tether * (V(pos, index, j) - V(x0, index, j));
    }
    for (int i = 0; i < s->np; i++) {
        int d = s->pd[i], p = s->pp[i];
        for (int j = 0; j < 3; j++) {
            double sep = V(pos, d, j) - V(pos, p, j);
            V(force, d, j) = -k_drude * sep;
            V(force, p, j) = k_drude * sep - k_tether * (V(pos, p, j) - V(x0, p, j));
        }
    }
}

int tgo_run_harness(tgo_state* s, double* pos, double* vel, double* force, const double* x0,
                    double k_drude, double k_tether, int nsteps) {
    for (int it = 0; it < nsteps; it++) {
        int rc = tgo_step_begin(s, pos, vel, force);
        if (rc != TGO_OK) return rc;
        tgo_harness_force(s, pos, x0, k_drude, k_tether, force);   /* Ref :384 ; Cu :380 call-out */
        tgo_step_end(s, vel, force);
    }
    return TGO_OK;
}

/* ------------------------------------------------------------------ */
/* Harness call-outs of the constrained path (not from the reference)  */
/* ------------------------------------------------------------------ */
void tgo_set_clusters(tgo_state* s, int n, const int* atoms, const int* ncons, const int* pairs, const double* dist) {
    free(s->cl_atoms); free(s->cl_ncons); free(s->cl_pairs); free(s->cl_dist);
    s->ncl = n;
    s->cl_atoms = xcalloc((size_t)n * 4, sizeof(int)); memcpy(s->cl_atoms, atoms, sizeof(int) * 4 * n);
    s->cl_ncons = xcalloc(n, sizeof(int)); memcpy(s->cl_ncons, ncons, sizeof(int) * n);
    s->cl_pairs = xcalloc((size_t)n * 12, sizeof(int)); memcpy(s->cl_pairs, pairs, sizeof(int) * 12 * n);
    s->cl_dist = xcalloc((size_t)n * 6, sizeof(double)); memcpy(s->cl_dist, dist, sizeof(double) * 6 * n);
}

#define SHAKE_MAX_ITER 500

int tgo_shake_positions(const tgo_state* s, const double* pos, double* delta, double tol) {
    for (int c = 0; c < s->ncl; c++) {
        const int* at = s->cl_atoms + 4 * c;
        const int nc = s->cl_ncons[c];
        int iter = 0, converged = 0;
        while (!converged && iter++ < SHAKE_MAX_ITER) {
            converged = 1;
            for (int k = 0; k < nc; k++) {
                const int i = at[s->cl_pairs[12 * c + 2 * k]], j = at[s->cl_pairs[12 * c + 2 * k + 1]];
                const double d = s->cl_dist[6 * c + k], d2 = d * d;
                double r[3], sv[3];
                for (int a = 0; a < 3; a++) {
                    r[a] = V(pos, i, a) - V(pos, j, a);
                    sv[a] = r[a] + (V(delta, i, a) - V(delta, j, a));
                }
                const double diff = d2 - dot3(sv, sv);
                if (fabs(diff) > 2.0 * tol * d2) {
                    converged = 0;
                    const double g = diff / (2.0 * dot3(r, sv) * (s->inv_mass[i] + s->inv_mass[j]));
                    for (int a = 0; a < 3; a++) {
                        V(delta, i, a) += g * s->inv_mass[i] * r[a];
                        V(delta, j, a) -= g * s->inv_mass[j] * r[a];
                    }
                }
            }
        }
        if (!converged) return fail(TGO_ERR_ARG, "SHAKE did not converge");
    }
    return TGO_OK;
}

int tgo_shake_velocities(const tgo_state* s, const double* pos, double* vel, double tol) {
    for (int c = 0; c < s->ncl; c++) {
        const int* at = s->cl_atoms + 4 * c;
        const int nc = s->cl_ncons[c];
        int iter = 0, converged = 0;
        while (!converged && iter++ < SHAKE_MAX_ITER) {
            converged = 1;
            for (int k = 0; k < nc; k++) {
                const int i = at[s->cl_pairs[12 * c + 2 * k]], j = at[s->cl_pairs[12 * c + 2 * k + 1]];
                double r[3], vr[3];
                for (int a = 0; a < 3; a++) { r[a] = V(pos, i, a) - V(pos, j, a); vr[a] = V(vel, i, a) - V(vel, j, a); }
                const double r2 = dot3(r, r), rv = dot3(r, vr);
                const double g = rv / (r2 * (s->inv_mass[i] + s->inv_mass[j]));
                if (fabs(g) * (s->inv_mass[i] + s->inv_mass[j]) > tol) {      /* relative bond-velocity rate above tol (1/ps) */
                    converged = 0;
                    for (int a = 0; a < 3; a++) {
                        V(vel, i, a) -= g * s->inv_mass[i] * r[a];
                        V(vel, j, a) += g * s->inv_mass[j] * r[a];
                    }
                }
            }
        }
        if (!converged) return fail(TGO_ERR_ARG, "velocity SHAKE did not converge");
    }
    return TGO_OK;
}

void tgo_set_virtual_sites(tgo_state* s, int n, const int* atoms, const double* w) {
    free(s->vs_atoms); free(s->vs_w);
    s->nvs = n;
    s->vs_atoms = xcalloc((size_t)n * 4, sizeof(int)); memcpy(s->vs_atoms, atoms, sizeof(int) * 4 * n);
    s->vs_w = xcalloc((size_t)n * 3, sizeof(double)); memcpy(s->vs_w, w, sizeof(double) * 3 * n);
}

void tgo_virtual_sites(const tgo_state* s, double* pos) {
    for (int k = 0; k < s->nvs; k++) {
        const int* a = s->vs_atoms + 4 * k;
        const double* w = s->vs_w + 3 * k;
        for (int j = 0; j < 3; j++)
            V(pos, a[0], j) = w[0] * V(pos, a[1], j) + w[1] * V(pos, a[2], j) + w[2] * V(pos, a[3], j);
    }
}

int tgo_run_harness_constrained(tgo_state* s, double* pos, double* vel, double* force, const double* x0,
                                double k_drude, double k_tether, double tol, int nsteps) {
    double* delta = xcalloc((size_t)s->n * 3, sizeof(double));
    int rc = TGO_OK;
    for (int it = 0; it < nsteps && rc == TGO_OK; it++) {
        tgo_propagate_nhc(s, vel, NULL, NULL);              /* Ref :231 ; Cu :336-353 */
        tgo_half_kick(s, vel, force);                       /* Ref :239 ; Cu :356-360 */
        const double dt = s->dt;
        for (int i = 0; i < s->n; i++)                      /* Ref :256-258 xPrime - pos ; K :323, :361-362 posDelta */
            for (int j = 0; j < 3; j++) V(delta, i, j) = (s->inv_mass[i] != 0.0) ? V(vel, i, j) * dt : 0.0;
        rc = tgo_shake_positions(s, pos, delta, tol);       /* Ref :268 ; Cu :363 */
        if (rc != TGO_OK) break;
        const double dtInv = 1.0 / dt;
        for (int i = 0; i < s->n; i++) {                    /* Ref :278-284 ; K :440-455 */
            if (s->inv_mass[i] != 0.0) {
                for (int j = 0; j < 3; j++) {
                    V(pos, i, j) += V(delta, i, j);
                    V(vel, i, j) = dtInv * V(delta, i, j);
                }
            }
        }
        rc = tgo_hardwall(s, pos, vel);                     /* Ref :298-363 ; Cu :372-376 */
        if (rc != TGO_OK) break;
        tgo_virtual_sites(s, pos);                          /* Ref :373 ; Cu :377 */
        tgo_harness_force(s, pos, x0, k_drude, k_tether, force);   /* Ref :384 ; Cu :380 */
        tgo_half_kick(s, vel, force);                       /* Ref :394 ; Cu :384-388 */
        if (s->mode == TGO_MODE_TGNH) {                     /* Cu :391 only: the Reference platform has no velocity stage */
            rc = tgo_shake_velocities(s, pos, vel, tol);
            if (rc != TGO_OK) break;
        }
        tgo_propagate_nhc(s, vel, NULL, NULL);              /* Ref :406 ; Cu :394-402 */
        s->time += s->dt;
        s->step_count++;
    }
    free(delta);
    return rc;
}

double tgo_time(const tgo_state* s) { return s->time; }
long tgo_step_count(const tgo_state* s) { return s->step_count; }
