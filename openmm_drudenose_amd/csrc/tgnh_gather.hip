// tgnh_gather.hip -- the step by GLOBAL INDEX: the slow path for topologies the tiled kernels cannot hold.
//
// tgnh_kernels.hip cuts the slot range into tiles inside which a Drude partner and the molecular centre of mass are on-chip
// look-ups, and keeps the kinetic-energy bins of <= 32 temperature groups in registers / per-wavefront LDS rows.  The reference
// has none of those limits: its kernels gather by arbitrary index (K :171-186 pairParticles, :123 particleResId) and size their
// bins by G + 2 (K :138-200, Cu :157).  So whatever tgnh_create cannot tile -- a Drude particle more than a tile away from its
// parent, pairs overlapping so densely in a long molecule that no cut between two pairs lies within a tile's reach, more than 32
// temperature groups -- steps through the kernels of this file instead: the reference's own un-fused form, every look-up a global
// load.  Work items are K's (a normal particle, a Drude pair, a residue) but walked in PARTICLE order: thread i looks at particle
// i, does a normal particle's work, or -- on the Drude particle -- its pair's, or nothing on a parent, so that a wavefront's loads
// and stores are consecutive wherever the topology is (K's three index lists walk the arrays two or three times, each with
// holes: twice the memory traffic at 5 M slots), and a thread asks for its own particle's data BEFORE it knows the particle's
// role (a parent's loads hit its neighbours' lines and are dropped): no load but a pair's second member waits for an index word.
// It exists so that nothing the reference accepts is refused (still refused:
// what the reference itself cannot run -- a massless pair member, Ref :132; a molecule without mass under the COM group,
// K :86-104; dualNH without a pair, Ref :181).
//
//   gather_com_kernel    K :82-113   calcCOMVelocities            2^k lanes per residue (k from the mean residue size)
//   gather_ke_kernel     K :138-200  computeNormalizedKineticEnergies (with K :119-133 normalizeVelocities folded in); Ref :439-460
//   gather_rowsum_kernel K :202-242  sumNormalizedKineticEnergies   (more than 34 thermostats; else chain_kernel's own sum)
//   gather_chain_kernel  Cu :559-642                               (more than 34 thermostats; else chain_kernel)
//   gather_update_kernel K :249-301, :307-365, :435-466, :471-574 ; Ref :516-584, :253-363   rescale / kick / drift / posDelta / move / hard wall
//
// Sums: fp64, no atomics, fixed order (per-wavefront bins filled in an order that depends on the data only, then wavefronts and
// work-groups in index order): reproducible bit for bit, like the tiled path's.
#include "tgnh_tile_device.h"

namespace tgnh {

// ---------------------------------------------------------------------------
// centre-of-mass velocity of every residue: (sum m v / sum m, w = 1 / sum m)   K :82-113
// kick: of the velocities after the half kick v + (dt/2) F / m
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void gather_com_kernel(const GatherArgs a) {
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::mixed4 mixed4;
    const mixed4* __restrict__ velm = reinterpret_cast<const mixed4*>(a.velm);
    mixed4* __restrict__ com = reinterpret_cast<mixed4*>(a.com);
    const int W = a.com_lanes;                                   // lanes per residue: a power of two <= 64 (launch_gather_com)
    const int sub = threadIdx.x & (W - 1);
    const long long team = ((long long)blockIdx.x * BLOCK + threadIdx.x) / W, nteams = (long long)gridDim.x * BLOCK / W;
    const mixed fscale = (mixed)(0.5 * a.dt / 4294967296.0);     // Cu :295
    const long long trips = (a.n_res + nteams - 1) / nteams;     // (every lane takes every trip: the sums below are wavefront-wide exchanges)
    for (long long k = 0; k < trips; k++) {
        const long long r = team + k * nteams;
        const int2 rt = r < a.n_res ? a.res_table[r] : make_int2(0, 0);      // (count, first particle): contiguous molecules, Cu :121-125
        double sx = 0, sy = 0, sz = 0, sm = 0;
        for (int j = sub; j < rt.x; j += W) {
            const int i = rt.y + j;
            if (i >= a.n) break;                                 // (a residue in several runs: K's walk of `count` particles from the last run's start may leave the array; it stops at its end here)
            const mixed4 v = velm[i];
            if (v.w != 0) {                                      // K :92
                mixed vx = v.x, vy = v.y, vz = v.z;
                if (a.kick_com) {
                    const mixed c = fscale * v.w;
                    vx += c * force_as(a.force[i], (mixed)0); vy += c * force_as(a.force[i + a.padded], (mixed)0); vz += c * force_as(a.force[i + 2 * a.padded], (mixed)0);
                }
                const mixed m = rcp_(v.w);
                sx += (double)(vx * m); sy += (double)(vy * m); sz += (double)(vz * m); sm += (double)m;
            }
        }
        for (int off = W >> 1; off; off >>= 1) {                 // butterfly over the residue's lanes: a fixed order
            sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sz += __shfl_xor(sz, off, 64); sm += __shfl_xor(sm, off, 64);
        }
        if (sub == 0 && r < a.n_res) {
            const double wi = 1.0 / sm;                          // K :101 (a molecule without mass: refused at create)
            com[r] = mk4((mixed)(sx * wi), (mixed)(sy * wi), (mixed)(sz * wi), (mixed)wi);
        }
    }
}

// ---------------------------------------------------------------------------
// kinetic-energy bins.  Work items, as K's three loops: residues (bin G: M v_com^2, K :152-158), normal particles (bin of their
// group: m |v - v_com|^2, K :161-168; Ref :443-448), pairs (bin of the DRUDE particle's group: (m1 + m2) |cm|^2, bin G + 1:
// mu |v2 - v1|^2, K :171-186; Ref :449-460), each particle relative to its OWN residue's centre of mass (K :123-129).
// Items 0 .. n-1 are the particles in index order (a pair is its Drude particle's item, a parent's item is empty), then the residues.
// One row of partial sums per work-group.  Group bins: one row of NT doubles per wavefront in LDS; per batch of 64 items a
// wavefront adds, group by group in the order the groups first appear in the batch, the 64-lane sum of that group's values.
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void gather_ke_kernel(const GatherArgs a) {

    typedef typename Prec<PREC>::mixed4 mixed4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* const bins = reinterpret_cast<double*>(smem);        // [BLOCK / 64][NT]
    const mixed4* __restrict__ velm = reinterpret_cast<const mixed4*>(a.velm);
    const mixed4* __restrict__ com = reinterpret_cast<const mixed4*>(a.com);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NT = a.NT, G = a.G;
    double* const wbins = bins + (size_t)wv * NT;
    for (int b = lane; b < NT; b += 64) wbins[b] = 0.0;
    double ke_com = 0.0, ke_drude = 0.0;
    const long long n_items = (long long)a.n + (a.use_com ? a.n_res : 0);
    for (long long base = (long long)blockIdx.x * BLOCK + (tid & ~63); base < n_items; base += (long long)gridDim.x * BLOCK) {   // (wavefront-uniform trips)
        const long long it = base + lane;
        bool has = false;
        int g = 0;
        double val = 0.0;
        // the thread's own particle, asked for before its partner word is back (a parent's loads hit the lines of its neighbours;
        // nothing of them is used): one dependent load level less
        const bool part = it < a.n;
        const int p = part ? (int)it : 0;
        const int pj = part ? a.partner[p] : 0;                  // partner | is-Drude << 31 ; -1: in no pair
        const mixed4 v1 = velm[p];
        const int g1 = a.group[p];
        double c1x = 0, c1y = 0, c1z = 0;
        if (a.use_com) { const mixed4 c = com[a.resid[p]]; c1x = c.x; c1y = c.y; c1z = c.z; }
        if (part && pj == -1) {                                  // K :161-168
            if (v1.w != 0) {
                const double rx = v1.x - c1x, ry = v1.y - c1y, rz = v1.z - c1z;
                val = (rx * rx + ry * ry + rz * rz) * (double)rcp_(v1.w);
                g = g1; has = true;
            }
        } else if (part && pj < 0) {                             // K :171-186 (the Drude particle's item)
            const int q = pj & 0x7fffffff;
            const mixed4 v2 = velm[q];
            double c2x = 0, c2y = 0, c2z = 0;
            if (a.use_com) { const mixed4 c2 = com[a.resid[q]]; c2x = c2.x; c2y = c2.y; c2z = c2.z; }
            const double r1x = v1.x - c1x, r1y = v1.y - c1y, r1z = v1.z - c1z;
            const double r2x = v2.x - c2x, r2y = v2.y - c2y, r2z = v2.z - c2z;
            const double mass1 = rcp_(v1.w), mass2 = rcp_(v2.w);
            const double invTot = rcp_(mass1 + mass2);
            const double m1f = invTot * mass1, m2f = invTot * mass2;
            const double cmx = r1x * m1f + r2x * m2f, cmy = r1y * m1f + r2y * m2f, cmz = r1z * m1f + r2z * m2f;
            const double rlx = r2x - r1x, rly = r2y - r1y, rlz = r2z - r1z;
            val = (cmx * cmx + cmy * cmy + cmz * cmz) * (mass1 + mass2);
            ke_drude += (rlx * rlx + rly * rly + rlz * rlz) * (mass1 * mass2 * invTot);      // reduced mass = 1 / invReducedMass (K :178, :185)
            g = g1; has = true;
        } else if (it >= a.n && it < n_items) {                  // K :152-158
            const mixed4 c = com[it - a.n];
            ke_com += ((double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z) / (double)c.w;
        }
        unsigned long long rem = __ballot(has);
        while (rem) {                                            // one round per distinct group in this batch, in order of first appearance
            const int src = __ffsll((long long)rem) - 1;
            const int g0 = __shfl(g, src, 64);
            const bool mine = has && g == g0;
            const double sg = wave_sum(mine ? val : 0.0);
            if (lane == 0) wbins[g0] += sg;
            rem &= ~__ballot(mine);
        }
    }
    ke_com = wave_sum(ke_com);
    ke_drude = wave_sum(ke_drude);
    if (lane == 0) { wbins[G] += ke_com; wbins[G + 1] += ke_drude; }
    __syncthreads();
    for (int b = tid; b < NT; b += BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) s += bins[(size_t)w * NT + b];     // fixed order
        a.partials[(size_t)blockIdx.x * NT + b] = s;
    }
}

// rows -> ke_red, column by column in row order (more than 34 thermostats: chain_kernel's sum keeps its columns in registers)
__global__ __launch_bounds__(BLOCK) void gather_rowsum_kernel(const double* __restrict__ partials, const int nrows, const int NT, double* __restrict__ ke_red) {
    const int b = (int)(blockIdx.x * BLOCK + threadIdx.x);
    if (b >= NT) return;
    double s = 0.0;
    for (int r = 0; r < nrows; r++) s += partials[(size_t)r * NT + b];
    ke_red[b] = s;
}

// ---------------------------------------------------------------------------
// the chain for more than 34 thermostats (TGNH mode; Cu :559-642): a thread per thermostat, the arithmetic of chain_kernel
// (run_tgnh: links in registers up to 4, else in a scratch row of 4 C + 1 doubles per thermostat -- global memory here)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void gather_chain_kernel(const ChainArgs a, double* scratch) {
    __shared__ double sred[BLOCK / 64];
    const ChainLayout& L = a.L;
    const int NT = L.NT, tid = threadIdx.x;
    double* st = a.st;
    double part = 0.0;
    for (int itg = tid; itg < NT; itg += BLOCK) {
        const double ke = st[L.off_ke_red + itg];
        if (a.status && ke != ke) atomicOr(a.status, 16u);       // (chain_prologue's check)
        part += ke;
        switch (L.C) {
            case 1: run_tgnh<1>(a, st, st, true, nullptr, itg, nullptr, ke); break;
            case 2: run_tgnh<2>(a, st, st, true, nullptr, itg, nullptr, ke); break;
            case 3: run_tgnh<3>(a, st, st, true, nullptr, itg, nullptr, ke); break;
            case 4: run_tgnh<4>(a, st, st, true, nullptr, itg, nullptr, ke); break;
            default: run_tgnh<0>(a, st, st, true, nullptr, itg, scratch, ke); break;
        }
    }
    part = wave_sum(part);                                       // Cu :493-497: KESum = 1/2 sum of the bins
    if ((tid & 63) == 0) sred[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < BLOCK / 64; w++) s += sred[w];
        st[L.off_kesum] = 0.5 * s;
    }
}

// ---------------------------------------------------------------------------
// rescale / half kick / drift / posDelta / move / hard wall, by index.  A work item is a normal particle or a PAIR (both
// members in one thread, as K: a pair's rescale reads both old velocities and writes both new ones); thread i takes particle
// i's item -- its own if it is in no pair, its pair's if it is the Drude particle, none if it is a parent.
// Order of the operations: tile_body's (A6, A8-move, A7, A8-drift, posDelta, A10).
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void gather_update_kernel(const GatherArgs a) {
    typedef typename Prec<PREC>::real real;
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed4 mixed4;
    mixed4* __restrict__ velm = reinterpret_cast<mixed4*>(a.velm);
    real4* __restrict__ posq = reinterpret_cast<real4*>(a.posq);
    float4* __restrict__ pcorr = reinterpret_cast<float4*>(a.posq_corr);
    mixed4* __restrict__ pdelta = reinterpret_cast<mixed4*>(a.pos_delta);
    const mixed4* __restrict__ com = reinterpret_cast<const mixed4*>(a.com);
    const int ops = a.ops, G = a.G;
    const bool do_scale = ops & OP_SCALE, do_kick = ops & OP_KICK, do_drift = ops & OP_DRIFT, do_pd = ops & OP_POSDELTA;
    const bool do_move = ops & OP_MOVE, do_prekick = ops & OP_PREKICK;
    const bool pos = do_drift || do_move, need_f = do_kick || do_prekick, hardwall = pos && a.hardwall != 0;
    const bool vel_w = do_scale || do_kick || do_move || do_prekick || hardwall;
    const mixed dt = (mixed)a.dt, fscale = (mixed)(0.5 * a.dt / 4294967296.0);       // Cu :295
    const double invStep = 1.0 / a.dt;                                            // K :436
    const mixed s_com = do_scale ? (mixed)a.scale[G] : (mixed)1, s_drude = do_scale ? (mixed)a.scale[G + 1] : (mixed)1;

    struct P { mixed4 v; mixed fx, fy, fz, px, py, pz; real pq; mixed4 pd; mixed cx, cy, cz; };
    auto load = [&](const int i, P& p) {
        p.v = velm[i];
        p.fx = p.fy = p.fz = 0; p.px = p.py = p.pz = 0; p.pq = 0; p.cx = p.cy = p.cz = 0;
        if (need_f) { p.fx = force_as(a.force[i], (mixed)0); p.fy = force_as(a.force[i + a.padded], (mixed)0); p.fz = force_as(a.force[i + 2 * a.padded], (mixed)0); }
        if (pos) {
            const real4 q = posq[i];
            p.px = q.x; p.py = q.y; p.pz = q.z; p.pq = q.w;
            if (PREC == TGNH_PREC_MIXED) { const float4 c = pcorr[i]; p.px += (mixed)c.x; p.py += (mixed)c.y; p.pz += (mixed)c.z; }   // K :443-445
        }
        if (do_move) p.pd = pdelta[i];
        if (do_scale && a.use_com) { const mixed4 c = com[a.resid[i]]; p.cx = c.x; p.cy = c.y; p.cz = c.z; }
    };
    auto kick = [&](P& p) {                                      // A7, per particle (tile_body; = Ref :560, :577-582)
        if (p.v.w != 0) {
            const mixed c = fscale * p.v.w;
            p.v.x += c * p.fx; p.v.y += c * p.fy; p.v.z += c * p.fz;
        }
    };
    auto after_scale = [&](P& p) {                               // move, kick, drift (one particle)
        if (do_move && p.v.w != 0) {                             // K :435-466
            p.px += p.pd.x; p.py += p.pd.y; p.pz += p.pd.z;
            p.v.x = (mixed)(invStep * p.pd.x); p.v.y = (mixed)(invStep * p.pd.y); p.v.z = (mixed)(invStep * p.pd.z);
        }
        if (do_kick) kick(p);
        if (do_drift && p.v.w != 0) { p.px += dt * p.v.x; p.py += dt * p.v.y; p.pz += dt * p.v.z; }     // Ref :253-258
    };
    auto store = [&](const int i, const P& p) {
        if (vel_w) velm[i] = p.v;
        if (do_pd) {                                             // K :322-324, :360-363
            const bool mv = p.v.w != 0;
            pdelta[i] = mk4(mv ? dt * p.v.x : (mixed)0, mv ? dt * p.v.y : (mixed)0, mv ? dt * p.v.z : (mixed)0, (mixed)0);
        }
        if (pos) {
            if (PREC == TGNH_PREC_MIXED) {                       // K :457-458
                const float hx = (float)p.px, hy = (float)p.py, hz = (float)p.pz;
                posq[i] = mk4((real)hx, (real)hy, (real)hz, p.pq);
                pcorr[i] = make_float4((float)(p.px - hx), (float)(p.py - hy), (float)(p.pz - hz), 0.0f);
            } else {
                posq[i] = mk4((real)p.px, (real)p.py, (real)p.pz, p.pq);
            }
        }
    };

    for (long long it = (long long)blockIdx.x * BLOCK + threadIdx.x; it < a.n; it += (long long)gridDim.x * BLOCK) {
        const int i = (int)it;
        const int pj = a.partner[i];                             // partner | is-Drude << 31 ; -1: in no pair
        P p1, p2;
        load(i, p1);                                             // the thread's own particle -- one instruction stream for both kinds of item, and a parent's too
                                                                 // (nothing of it is used: its lines are its neighbours'): no load of it waits for the partner word
        if (pj >= 0) continue;                                   // a parent: its Drude particle's thread does the pair
        const bool pair = pj != -1;
        const int2 pr = make_int2(i, pair ? pj & 0x7fffffff : i);   // (Drude particle, parent): K's particles.x, .y
        const mixed s_g = do_scale ? (mixed)a.scale[a.group[i]] : (mixed)1;      // (a pair's group is its Drude particle's, K :270)
        if (!pair) {
            P& p = p1;
            if (do_prekick) kick(p);
            if (do_scale && p.v.w != 0) {                        // K :260-265
                const mixed rx = p.v.x - p.cx, ry = p.v.y - p.cy, rz = p.v.z - p.cz;
                p.v.x = s_g * rx + s_com * (p.v.x - rx);
                p.v.y = s_g * ry + s_com * (p.v.y - ry);
                p.v.z = s_g * rz + s_com * (p.v.z - rz);
            }
            after_scale(p);
        } else {
            load(pr.y, p2);
            if (do_prekick) { kick(p1); kick(p2); }
            const mixed mass1 = rcp_(p1.v.w), mass2 = rcp_(p2.v.w);      // (massless pair members: refused at create)
            const mixed invTot = rcp_(mass1 + mass2);
            if (do_scale) {                                      // K :270-300
                const mixed m1f = invTot * mass1, m2f = invTot * mass2;
                const mixed r1x = p1.v.x - p1.cx, r1y = p1.v.y - p1.cy, r1z = p1.v.z - p1.cz;
                const mixed r2x = p2.v.x - p2.cx, r2y = p2.v.y - p2.cy, r2z = p2.v.z - p2.cz;
                const mixed cmx = s_g * (r1x * m1f + r2x * m2f), cmy = s_g * (r1y * m1f + r2y * m2f), cmz = s_g * (r1z * m1f + r2z * m2f);
                const mixed rlx = s_drude * (r2x - r1x), rly = s_drude * (r2y - r1y), rlz = s_drude * (r2z - r1z);
                p1.v.x = cmx - rlx * m2f + s_com * (p1.v.x - r1x);
                p1.v.y = cmy - rly * m2f + s_com * (p1.v.y - r1y);
                p1.v.z = cmz - rlz * m2f + s_com * (p1.v.z - r1z);
                p2.v.x = cmx + rlx * m1f + s_com * (p2.v.x - r2x);
                p2.v.y = cmy + rly * m1f + s_com * (p2.v.y - r2y);
                p2.v.z = cmz + rlz * m1f + s_com * (p2.v.z - r2z);
            }
            after_scale(p1); after_scale(p2);
            if (hardwall) {                                      // K :471-574 ; Ref :298-363 (tile_body's arithmetic, both members here)
                const mixed maxd = (mixed)a.max_dist, hws = (mixed)a.hw_scale;
                const mixed dx = p1.px - p2.px, dy = p1.py - p2.py, dz = p1.pz - p2.pz;          // Drude - parent (K :487)
                const mixed d2 = dx * dx + dy * dy + dz * dz;
                if (d2 > maxd * maxd) {
                    const mixed r = sqrt_(d2);
                    const mixed rInv = rcp_(r);
                    if (rInv * maxd < (mixed)0.5) atomicOr(a.status, 1u);                         // Ref :311-312
                    const mixed bx = dx * rInv, by = dy * rInv, bz = dz * rInv;
                    const mixed deltaR = r - maxd;
                    mixed deltaT = dt;
                    mixed dotvr1 = p1.v.x * bx + p1.v.y * by + p1.v.z * bz;
                    const mixed vp1x = p1.v.x - bx * dotvr1, vp1y = p1.v.y - by * dotvr1, vp1z = p1.v.z - bz * dotvr1;
                    mixed dotvr2 = p2.v.x * bx + p2.v.y * by + p2.v.z * bz;
                    const mixed vp2x = p2.v.x - bx * dotvr2, vp2y = p2.v.y - by * dotvr2, vp2z = p2.v.z - bz * dotvr2;
                    const mixed vbCMass = (mass1 * dotvr1 + mass2 * dotvr2) * invTot;
                    dotvr1 -= vbCMass;
                    dotvr2 -= vbCMass;
                    if (dotvr1 != dotvr2) deltaT = deltaR / abs_(dotvr1 - dotvr2);
                    if (deltaT > dt) deltaT = dt;
                    const mixed vBond = hws / sqrt_(mass1);
                    dotvr1 = -dotvr1 * vBond * mass2 * invTot / abs_(dotvr1);
                    dotvr2 = -dotvr2 * vBond * mass1 * invTot / abs_(dotvr2);
                    const mixed dr1 = -deltaR * mass2 * invTot + deltaT * dotvr1;
                    const mixed dr2 = deltaR * mass1 * invTot + deltaT * dotvr2;
                    dotvr1 += vbCMass;
                    dotvr2 += vbCMass;
                    p1.px += bx * dr1; p1.py += by * dr1; p1.pz += bz * dr1;
                    p1.v.x = vp1x + bx * dotvr1; p1.v.y = vp1y + by * dotvr1; p1.v.z = vp1z + bz * dotvr1;
                    p2.px += bx * dr2; p2.py += by * dr2; p2.pz += bz * dr2;
                    p2.v.x = vp2x + bx * dotvr2; p2.v.y = vp2y + by * dotvr2; p2.v.z = vp2z + bz * dotvr2;
                }
            }
            store(pr.y, p2);
        }
        store(i, p1);
    }
}

// ---------------------------------------------------------------------------
// harness force by index (bench / test workload, not part of the reference): the forces of force_kernel -- tether of the
// massive non-Drude sites, Drude spring -- with the partner from an index array instead of the tiled path's per-slot offset
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void gather_force_kernel(const GatherArgs a, const void* x0_, long long* force, const double k_drude, const double k_tether) {
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed mixed;
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    const real4* __restrict__ x0 = reinterpret_cast<const real4*>(x0_);
    const mixed kd = (mixed)k_drude, kt = (mixed)k_tether;
    auto position = [&](const int i, mixed& x, mixed& y, mixed& z) {
        const real4 p = posq[i];
        x = p.x; y = p.y; z = p.z;
        if (PREC == TGNH_PREC_MIXED) { const float4 c = pcorr[i]; x += (mixed)c.x; y += (mixed)c.y; z += (mixed)c.z; }
    };
    for (long long it = (long long)blockIdx.x * BLOCK + threadIdx.x; it < a.n; it += (long long)gridDim.x * BLOCK) {
        const int i = (int)it;
        const int pj = a.partner[i];
        mixed x, y, z;
        position(i, x, y, z);
        mixed fx = 0, fy = 0, fz = 0;
        const bool paired = pj != -1, is_d = paired && (pj < 0);
        if (!is_d) {                                             // tether of massive non-Drude sites
            const real4 s = x0[i];
            if (s.w != 0) { fx = -kt * (x - (mixed)s.x); fy = -kt * (y - (mixed)s.y); fz = -kt * (z - (mixed)s.z); }
        }
        if (paired) {                                            // Drude spring: separation Drude - parent; -k sep on the Drude, +k sep on the parent
            mixed ox, oy, oz;
            position(pj & 0x7fffffff, ox, oy, oz);
            const mixed sgn = is_d ? (mixed)-1 : (mixed)1;
            const mixed sx = is_d ? x - ox : ox - x, sy = is_d ? y - oy : oy - y, sz = is_d ? z - oz : oz - z;
            fx += sgn * kd * sx; fy += sgn * kd * sy; fz += sgn * kd * sz;
        }
        force[i] = (long long)(fx * (mixed)4294967296.0);
        force[i + a.padded] = (long long)(fy * (mixed)4294967296.0);
        force[i + 2 * a.padded] = (long long)(fz * (mixed)4294967296.0);
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static int gather_grid(const long long items, const int cap) {
    long long g = (items + BLOCK - 1) / BLOCK;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

#define GATHER_BY_PREC(kernel, grid, lds, s, ...)                                                                          \
    switch (precision) {                                                                                                   \
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((kernel<TGNH_PREC_SINGLE>), dim3(grid), dim3(BLOCK), lds, s, __VA_ARGS__); break; \
        case TGNH_PREC_MIXED: TGNH_LAUNCH((kernel<TGNH_PREC_MIXED>), dim3(grid), dim3(BLOCK), lds, s, __VA_ARGS__); break;   \
        case TGNH_PREC_DOUBLE: TGNH_LAUNCH((kernel<TGNH_PREC_DOUBLE>), dim3(grid), dim3(BLOCK), lds, s, __VA_ARGS__); break; \
        default: return hipErrorInvalidValue;                                                                              \
    }

hipError_t launch_gather_com(int precision, const GatherArgs& a, hipStream_t s) {
    const int grid = gather_grid((long long)a.n_res * a.com_lanes, 8192);
    GATHER_BY_PREC(gather_com_kernel, grid, 0, s, a)
    return hipGetLastError();
}
int gather_ke_grid(const GatherArgs& a) {
    return gather_grid((long long)a.n + (a.use_com ? a.n_res : 0), GATHER_KE_ROWS);
}
hipError_t launch_gather_ke(int precision, const GatherArgs& a, int grid, hipStream_t s) {
    const size_t lds = sizeof(double) * (BLOCK / 64) * (size_t)a.NT;
    if (lds > 64 * 1024) return hipErrorInvalidValue;                          // (tgnh_create refuses more groups than fit)
    GATHER_BY_PREC(gather_ke_kernel, grid, lds, s, a)
    return hipGetLastError();
}
hipError_t launch_gather_rowsum(const double* partials, int nrows, int NT, double* ke_red, hipStream_t s) {
    TGNH_LAUNCH(gather_rowsum_kernel, dim3((NT + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, partials, nrows, NT, ke_red);
    return hipGetLastError();
}
hipError_t launch_gather_chain(const ChainArgs& a, double* scratch, hipStream_t s) {
    TGNH_LAUNCH(gather_chain_kernel, dim3(1), dim3(BLOCK), 0, s, a, scratch);
    return hipGetLastError();
}
hipError_t launch_gather_update(int precision, const GatherArgs& a, hipStream_t s) {
    const int grid = gather_grid(a.n, 8192);
    GATHER_BY_PREC(gather_update_kernel, grid, 0, s, a)
    return hipGetLastError();
}
hipError_t launch_gather_force(int precision, const GatherArgs& a, const void* x0, long long* force, double k_drude, double k_tether, hipStream_t s) {
    const int grid = gather_grid(a.n, 8192);
    GATHER_BY_PREC(gather_force_kernel, grid, 0, s, a, x0, force, k_drude, k_tether)
    return hipGetLastError();
}

}  // namespace tgnh
