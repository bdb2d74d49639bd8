// tgnh_kernels.hip -- gfx950 (CDNA4) kernels of the DrudeTGNHIntegrator step.
//
// Design (DESIGN.md "Kernels"): the whole per-particle part of a thermostat or
// velocity-Verlet half step is ONE streaming pass of `tile_kernel`.  A 256-thread
// work-group (4 wavefronts x 64 lanes) owns a tile of <= 512 consecutive slots whose
// ends never cut a Drude pair or a molecule, loads it with one coalesced 16/32-byte
// access per lane, keeps the velocity image in LDS so the Drude partner and the
// molecular centre of mass are LDS look-ups, and leaves with fp64 per-group kinetic
// energy sums reduced over the 64 lanes (wave_sum) + one LDS hop.  Which of
// {rescale, half kick, drift, hard wall, KE} a launch performs is a compile-time mask,
// so e.g. rescale+kick+drift touches each array once.  The Nose-Hoover chains run
// on the device (chain_kernel, fp64), so a step has no host round trip.
// Shared device code (precision traits, 64-lane sums, the kinetic-energy bins, the meeting of the one-launch step, the per-tile
// work of the wave-tile step): tgnh_tile_device.h; the chain: tgnh_chain_device.h.  What these tiles cannot hold -- a Drude particle
// more than a tile from its parent, more than 32 temperature groups, residues in several runs -- steps through the reference's
// own un-fused kernels by global index instead: tgnh_gather.hip.
//
// Reference semantics followed (scychon/openmm_drudeNose):
//   K  = platforms/cuda/src/kernels/drudeTGNH.cu
//   Cu = platforms/cuda/src/CudaDrudeTGNHKernels.cpp
//   Ref= platforms/reference/src/ReferenceDrudeTGNHKernels.cpp
#include "tgnh_tile_device.h"

#ifdef TGNH_TRACE
TGNH_TRACE_READERS(tgnh_debug_read_trace, tgnh_debug_clear_trace)
extern "C" int tgnh_debug_read_chain_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnh::g_chain_trace), sizeof(unsigned long long) * 8);
}
#endif

namespace tgnh {


size_t tile_lds_bytes(int precision, int ops, bool hardwall, bool use_com) {
    (void)use_com;
    const size_t m4 = (precision == TGNH_PREC_SINGLE) ? 16 : 32;
    const bool hw = hardwall && (ops & (OP_DRIFT | OP_MOVE));
    size_t b = (size_t)(TBLOCK / 64) * (MAX_GROUPS + 2) * 8;          // KE reduction scratch
    if ((ops & (OP_SCALE | OP_KE)) || hw) b = m4 * (TILE_SLOTS + TILE_RES);   // sv + scom (fixed carve)
    if (hw) b += m4 * TILE_SLOTS;                                     // sx
    return b;
}

// ---------------------------------------------------------------------------
// tile_kernel
// ---------------------------------------------------------------------------
#ifndef TGNH_MINWAVES
#define TGNH_MINWAVES 1
#endif

// Raw register image of one tile's global loads.
template <int PREC> struct TileIn {
    int ts, te, rs, nres;
    typename Prec<PREC>::mixed4 v[SPT];
    uint32_t meta[SPT];
    long long fx[SPT], fy[SPT], fz[SPT];
    typename Prec<PREC>::real4 p[SPT];
    float4 c[SPT];
    typename Prec<PREC>::mixed4 pd[SPT];
    int2 rt;                 // this lane's molecule entry (lane r < nres): fetched with the tile, not after the first barrier
};

// Tiles of identical molecules (PATTERN_WORDS, tgnh_internal.h): a thread's slots sit at the same positions in every tile, so the
// words it forms for one tile of a pattern are the words of the next -- kept in registers, formed again when the pattern changes
// (a water box: once per launch; the pattern's lines would otherwise be fetched by every wavefront of the chip for every tile).
struct TilePattern {
    uint32_t pat = 0u;
    uint32_t word[SPT] = {};
};

// which arrays a pass touches
template <int OPS> struct OpsOf {
    static constexpr bool DO_SCALE = OPS & OP_SCALE, DO_KICK = OPS & OP_KICK, DO_DRIFT = OPS & OP_DRIFT;
    static constexpr bool DO_KE = OPS & OP_KE, DO_PD = OPS & OP_POSDELTA, DO_MOVE = OPS & OP_MOVE;
    // OP_PREKICK: the half kick a kick+KE pass of the previous step formed for its sums but did not store
    // (OP_NOSTORE) is applied first -- same force buffer, same expression, same bits (DESIGN.md "deferred kick")
    static constexpr bool DO_PREKICK = OPS & OP_PREKICK, NOSTORE = OPS & OP_NOSTORE;
    static constexpr bool NEED_F = DO_KICK || DO_PREKICK;
    static constexpr bool POS = DO_DRIFT || DO_MOVE;            // positions are read and written
    static constexpr bool VEL_W = (DO_SCALE || DO_KICK || DO_MOVE) && !NOSTORE;   // velocities are written
};

// issue the global loads of tile t for a pass with operations OPS.  HAVE != 0: `in` already holds what a pass with
// operations HAVE loaded for this very tile (velocities, index words, and its forces if it needed them): fetch the rest.
// PATTERN = false: always the per-slot words (tile_kernel's read-only KE passes, at their register budget of 5 work-groups per CU).
template <int PREC, int OPS, int HAVE = 0, bool PATTERN = true>
__device__ __forceinline__ void tile_load(const TileArgs& a, const int t, TileIn<PREC>& in, TilePattern& tp) {
    constexpr bool KEEP_VF = HAVE != 0;
    constexpr bool LOAD_F = OpsOf<OPS>::NEED_F && !(KEEP_VF && OpsOf<HAVE>::NEED_F);
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed4 mixed4;
    typedef OpsOf<OPS> O;
    const int tid = threadIdx.x;
    const mixed4* __restrict__ velm = reinterpret_cast<const mixed4*>(a.velm);
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    const mixed4* __restrict__ pdelta = reinterpret_cast<const mixed4*>(a.pos_delta);
    if (!KEEP_VF) {
        in.ts = a.tile_start[t]; in.te = a.tile_start[t + 1];
        in.rs = a.tile_res[t]; in.nres = a.tile_res[t + 1] - in.rs;
        if ((O::DO_SCALE || O::DO_KE) && a.use_com && tid < in.nres) in.rt = a.res_table[in.rs + tid];
    }
    // a tile of identical molecules: the slot's word from its position (TilePattern), no 4 B per slot from HBM
    const uint32_t pat = (KEEP_VF || !PATTERN) ? 0u : a.tile_pat[t];
    if (pat != 0u && pat != tp.pat) {                             // (work-group-uniform)
        const int period = (int)(pat & 255u), mols = (int)((pat >> 8) & 255u);
        const uint32_t* __restrict__ words = a.pattern + (size_t)(pat >> 16) * PATTERN_WORDS;
        const float rperiod = __builtin_amdgcn_rcpf((float)period);
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            const int pos = k * TBLOCK + tid;
            const int q = (int)(((float)pos + 0.5f) * rperiod);          // pos div period (pos < 512, period <= 64: never within rounding of an integer)
            tp.word[k] = words[pos - q * period] + (a.use_com ? (uint32_t)(q * mols) << 21 : 0u);
        }
        tp.pat = pat;
    }
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const int idx = in.ts + k * TBLOCK + tid;
        if (idx < in.te) {
            if (!KEEP_VF) {
                in.v[k] = velm[idx];
                if (pat == 0u) in.meta[k] = a.meta[idx];
                else in.meta[k] = tp.word[k];
            }
            if (LOAD_F) {
                in.fx[k] = a.force[idx];
                in.fy[k] = a.force[idx + a.padded];
                in.fz[k] = a.force[idx + 2 * a.padded];
            }
            if (O::POS) {
                in.p[k] = posq[idx];
                if (PREC == TGNH_PREC_MIXED) in.c[k] = pcorr[idx];       // K :443-445
            }
            if (O::DO_MOVE) in.pd[k] = pdelta[idx];
        } else if (!KEEP_VF) {
            in.v[k] = mk4((mixed)0, (mixed)0, (mixed)0, (mixed)0);       // w = 0: treated as massless, never stored
            in.meta[k] = 0u;
        }
    }
}

// One tile, loaded into `cur`, through the operations OPS (A3/A4, A6, A7, A8, A10).  Ends with the LDS images free.
// reuse_img (step_kernel): the velocity image and the COM table of this very tile are still in LDS from the pass before.
template <int PREC, int OPS, int GB>
__device__ __forceinline__ void tile_body(const TileArgs& a, TileEnv<PREC, GB>& e, const TileIn<PREC>& cur, const int trace_tile,
                                          const bool reuse_img = false) {
    typedef typename Prec<PREC>::real real;
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed4 mixed4;
    typedef OpsOf<OPS> O;
    typedef TileEnv<PREC, GB> E;
    constexpr bool DO_SCALE = O::DO_SCALE, DO_KICK = O::DO_KICK, DO_DRIFT = O::DO_DRIFT, DO_KE = O::DO_KE, DO_PD = O::DO_PD;
    constexpr bool DO_MOVE = O::DO_MOVE, DO_PREKICK = O::DO_PREKICK, NEED_F = O::NEED_F, POS = O::POS, VEL_W = O::VEL_W;
    (void)trace_tile;
    mixed4* const sv = e.sv; mixed4* const scom = e.scom; mixed4* const sx = e.sx;
    const double* const s_scale = e.s_scale;
    const int tid = e.tid, G = e.G;
    const bool use_com = e.use_com;
    const bool hardwall = POS && (a.hardwall != 0);
    const mixed dt = e.dt, fscale = e.fscale, s_com = e.s_com, s_drude = e.s_drude;
    double (&ke_g)[E::GBR] = e.ke_g;
    double& ke_com = e.ke_com; double& ke_drude = e.ke_drude;
    double* const wbins = e.wbins0 + (tid >> 6) * G;
    mixed4* __restrict__ velm = reinterpret_cast<mixed4*>(a.velm);
    real4* __restrict__ posq = reinterpret_cast<real4*>(a.posq);
    float4* __restrict__ pcorr = reinterpret_cast<float4*>(a.posq_corr);
    mixed4* __restrict__ pdelta = reinterpret_cast<mixed4*>(a.pos_delta);
    auto st_img = [&](mixed4* img, int i, const mixed4& u) { E::st_img(img, i, u); };
    auto ld_img = [&](const mixed4* img, int i) -> mixed4 { return E::ld_img(img, i); };
    (void)G; (void)s_com; (void)s_drude; (void)dt; (void)fscale; (void)wbins; (void)ke_com; (void)ke_drude; (void)ke_g;
    (void)posq; (void)pcorr; (void)pdelta; (void)velm; (void)s_scale;

    const int ts = cur.ts, te = cur.te;
    const int rs = cur.rs, nres = cur.nres;
    TRACE_WAIT(); TRACE(3 + 4 * trace_tile);

    mixed4 v[SPT];
    uint32_t meta[SPT];
    long long fx[SPT], fy[SPT], fz[SPT];
    mixed px[SPT], py[SPT], pz[SPT];
    real pq[SPT];
    mixed4 pd[SPT];
    bool ok[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        ok[k] = ts + k * TBLOCK + tid < te;
        v[k] = cur.v[k];
        meta[k] = cur.meta[k];
        if (NEED_F) { fx[k] = cur.fx[k]; fy[k] = cur.fy[k]; fz[k] = cur.fz[k]; }
        if (POS) {
            px[k] = cur.p[k].x; py[k] = cur.p[k].y; pz[k] = cur.p[k].z; pq[k] = cur.p[k].w;
            if (PREC == TGNH_PREC_MIXED) { px[k] += (mixed)cur.c[k].x; py[k] += (mixed)cur.c[k].y; pz[k] += (mixed)cur.c[k].z; }
        }
        if (DO_MOVE) pd[k] = cur.pd[k];
    }
    // One fp64 division per slot: the mass.  The LDS images carry it in .w (0 = massless), so the per-molecule walk
    // and the pair arithmetic multiply by masses instead of dividing by inverse masses again (K forms RECIP(w) in
    // every kernel; an fp64 reciprocal is ~15 VALU instructions and these launches are VALU-heavy at small sizes).
    mixed mass[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) mass[k] = v[k].w != 0 ? rcp_(v[k].w) : (mixed)0;
    if (DO_PREKICK) {                                            // the pending half kick (A7), as below
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            if (v[k].w != 0) {
                const mixed c = fscale * v[k].w;
                v[k].x += c * (mixed)fx[k];
                v[k].y += c * (mixed)fy[k];
                v[k].z += c * (mixed)fz[k];
            }
        }
    }
    auto img = [&](int k) { return mk4(v[k].x, v[k].y, v[k].z, mass[k]); };

    bool lds_read = false;   // some lane may still be reading sv/scom of this tile

    // ---------------- A6: rescale (K :249-301 ; Ref :516-541) ----------------
    if (DO_SCALE) {
      if (!reuse_img) {
#pragma unroll
        for (int k = 0; k < SPT; k++) st_img(sv, k * TBLOCK + tid, img(k));
        __syncthreads();
        if (use_com) {
            for (int r = tid; r < nres; r += TBLOCK) {            // K :86-111
                const int2 rt = r == tid ? cur.rt : a.res_table[rs + r];
                if (rt.x < 0) { scom[r] = reinterpret_cast<const mixed4*>(a.big_com)[-rt.x - 1]; continue; }   // molecule longer than a tile
                const int first = rt.y - ts;
                mixed cx = 0, cy = 0, cz = 0, cm = 0;
                for (int j = 0; j < rt.x; j++) {
                    const mixed4 u = ld_img(sv, first + j);
                    const mixed m = u.w;                       // mass (0 for massless sites)
                    cx += u.x * m; cy += u.y * m; cz += u.z * m; cm += m;
                }
                const mixed w = rcp_(cm);
                scom[r] = mk4(cx * w, cy * w, cz * w, w);
            }
            __syncthreads();
        }
      }
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            const uint32_t m = meta[k];
            const uint32_t role = m & 3u, g = (m >> 2) & 255u;
            mixed cx = 0, cy = 0, cz = 0;
            if (use_com) { const mixed4 c = scom[m >> 21]; cx = c.x; cy = c.y; cz = c.z; }
            const mixed s_g = (mixed)s_scale[g];
            if (role == ROLE_NORMAL) {
                if (v[k].w != 0) {                               // K :260-265
                    const mixed rx = v[k].x - cx, ry = v[k].y - cy, rz = v[k].z - cz;
                    v[k].x = s_g * rx + s_com * (v[k].x - rx);
                    v[k].y = s_g * ry + s_com * (v[k].y - ry);
                    v[k].z = s_g * rz + s_com * (v[k].z - rz);
                }
            } else {                                             // K :270-300
                // Written from the lane's own point of view (self s, partner p), which needs no role selects:
                // with cm = (r_s m_s + r_p m_p)/M and K's rel = r_parent - r_drude, both
                //   v_drude'  = s_g cm - s_D rel m_parent/M + s_COM v_com      (K :292-294)
                //   v_parent' = s_g cm + s_D rel m_drude/M  + s_COM v_com      (K :295-297)
                // read  v_s' = s_g cm + s_D (r_s - r_p) m_p/M + s_COM (v_s - r_s).
                const int pl = k * TBLOCK + tid + (int)((m >> 10) & 2047u) - 1024;
                const mixed4 u = ld_img(sv, pl);           // partner velocity, .w = partner mass
                const mixed rsx = v[k].x - cx, rsy = v[k].y - cy, rsz = v[k].z - cz;
                const mixed rpx = u.x - cx, rpy = u.y - cy, rpz = u.z - cz;
                const mixed invTot = rcp_(mass[k] + u.w);
                const mixed msf = invTot * mass[k], mpf = invTot * u.w;
                const mixed sdp = s_drude * mpf;
                v[k].x = s_g * (rsx * msf + rpx * mpf) + sdp * (rsx - rpx) + s_com * (v[k].x - rsx);
                v[k].y = s_g * (rsy * msf + rpy * mpf) + sdp * (rsy - rpy) + s_com * (v[k].y - rsy);
                v[k].z = s_g * (rsz * msf + rpz * mpf) + sdp * (rsz - rpz) + s_com * (v[k].z - rsz);
            }
        }
        lds_read = true;
    }

    TRACE(4 + 4 * trace_tile);
    // ---------------- A8 (constrained path): x += posDelta, v = posDelta/dt (K :435-466) ---
    if (DO_MOVE) {
        const double invStep = 1.0 / a.dt;                       // K :436
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            if (v[k].w != 0) {
                px[k] += pd[k].x; py[k] += pd[k].y; pz[k] += pd[k].z;
                v[k].x = (mixed)(invStep * pd[k].x);
                v[k].y = (mixed)(invStep * pd[k].y);
                v[k].z = (mixed)(invStep * pd[k].z);
            }
        }
    }

    // ---------------- A7: half kick (K :307-365 ; Ref :548-584) ----------------
    // Per-particle form v += (dt/2) F/m.  The reference writes the pair kick in
    // COM/relative coordinates; that is algebraically the same update
    // (tests/test_oracle.py::test_pair_kick_identity), so no partner access is needed here.
    if (DO_KICK) {
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            if (v[k].w != 0) {
                const mixed c = fscale * v[k].w;
                v[k].x += c * (mixed)fx[k];
                v[k].y += c * (mixed)fy[k];
                v[k].z += c * (mixed)fz[k];
            }
        }
    }

    // ---------------- A8: drift (Ref :253-258 ; K :322-324, :450-452) ----------------
    if (DO_DRIFT) {
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            if (v[k].w != 0) {
                px[k] += dt * v[k].x; py[k] += dt * v[k].y; pz[k] += dt * v[k].z;
            }
        }
    }
    if (DO_PD) {
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            const int idx = ts + k * TBLOCK + tid;
            if (ok[k]) {
                const bool mv = v[k].w != 0;
                pdelta[idx] = mk4(mv ? dt * v[k].x : (mixed)0, mv ? dt * v[k].y : (mixed)0, mv ? dt * v[k].z : (mixed)0, (mixed)0);
            }
        }
    }

    // ---------------- A10: hard wall (K :471-574 ; Ref :298-363) ----------------
    if (POS && hardwall) {
        if (lds_read) __syncthreads();
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            st_img(sv, k * TBLOCK + tid, img(k));
            st_img(sx, k * TBLOCK + tid, mk4(px[k], py[k], pz[k], (mixed)0));
        }
        __syncthreads();
        const mixed maxd = (mixed)a.max_dist, hws = (mixed)a.hw_scale;
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            const uint32_t m = meta[k];
            const uint32_t role = m & 3u;
            if (role != ROLE_NORMAL) {
                const int pl = k * TBLOCK + tid + (int)((m >> 10) & 2047u) - 1024;
                const mixed4 ux = ld_img(sx, pl);
                const mixed sxd = px[k] - ux.x, syd = py[k] - ux.y, szd = pz[k] - ux.z;     // self - partner
                const mixed d2 = sxd * sxd + syd * syd + szd * szd;
                if (d2 > maxd * maxd) {                           // r > max  <=>  rInv*max < 1 (K :490): the rest only for violators
                    const mixed4 uv = ld_img(sv, pl);
                    const bool is_d = role == ROLE_DRUDE;
                    const mixed4 vel1 = is_d ? v[k] : uv, vel2 = is_d ? uv : v[k];
                    const mixed dx = is_d ? sxd : -sxd, dy = is_d ? syd : -syd, dz = is_d ? szd : -szd;   // Drude - parent (K :487)
                    const mixed r = sqrt_(d2);
                    const mixed rInv = rcp_(r);
                    if (rInv * maxd < (mixed)0.5) atomicOr(a.status, 1u);     // Ref :311-312
                    const mixed bx = dx * rInv, by = dy * rInv, bz = dz * rInv;
                    const mixed mass1 = is_d ? mass[k] : uv.w, mass2 = is_d ? uv.w : mass[k];   // image .w = mass
                    const mixed deltaR = r - maxd;
                    mixed deltaT = dt;
                    mixed dotvr1 = vel1.x * bx + vel1.y * by + vel1.z * bz;
                    const mixed vp1x = vel1.x - bx * dotvr1, vp1y = vel1.y - by * dotvr1, vp1z = vel1.z - bz * dotvr1;
                    // K :527-571 (a massless parent, K :504-526, cannot occur: tgnh_create rejects massless pair members)
                    const mixed invTot = rcp_(mass1 + mass2);
                    mixed dotvr2 = vel2.x * bx + vel2.y * by + vel2.z * bz;
                    const mixed vp2x = vel2.x - bx * dotvr2, vp2y = vel2.y - by * dotvr2, vp2z = vel2.z - bz * dotvr2;
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
                    if (is_d) {
                        px[k] += bx * dr1; py[k] += by * dr1; pz[k] += bz * dr1;
                        v[k].x = vp1x + bx * dotvr1; v[k].y = vp1y + by * dotvr1; v[k].z = vp1z + bz * dotvr1;
                    } else {
                        px[k] += bx * dr2; py[k] += by * dr2; pz[k] += bz * dr2;
                        v[k].x = vp2x + bx * dotvr2; v[k].y = vp2y + by * dotvr2; v[k].z = vp2z + bz * dotvr2;
                    }
                }
            }
        }
        lds_read = true;
    }

    TRACE(5 + 4 * trace_tile);
    // ---------------- stores ----------------
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const int idx = ts + k * TBLOCK + tid;
        if (ok[k]) {
            if (VEL_W || (POS && hardwall)) velm[idx] = v[k];
            if (POS) {
                if (PREC == TGNH_PREC_MIXED) {                   // K :457-458
                    const float hx = (float)px[k], hy = (float)py[k], hz = (float)pz[k];
                    posq[idx] = mk4((real)hx, (real)hy, (real)hz, pq[k]);
                    pcorr[idx] = make_float4((float)(px[k] - hx), (float)(py[k] - hy), (float)(pz[k] - hz), 0.0f);
                } else {
                    posq[idx] = mk4((real)px[k], (real)py[k], (real)pz[k], pq[k]);
                }
            }
        }
    }

    // ---------------- A3/A4: kinetic energies (K :82-200 ; Ref :439-460) ----------------
    if (DO_KE) {
        if (lds_read) __syncthreads();
#pragma unroll
        for (int k = 0; k < SPT; k++) st_img(sv, k * TBLOCK + tid, img(k));
        __syncthreads();
        if (use_com) {
            for (int r = tid; r < nres; r += TBLOCK) {            // K :86-111, :152-158
                const int2 rt = r == tid ? cur.rt : a.res_table[rs + r];
                if (rt.x < 0) { scom[r] = reinterpret_cast<const mixed4*>(a.big_com)[-rt.x - 1]; continue; }   // its M v_com^2 comes from big_com_kernel
                const int first = rt.y - ts;
                mixed cx = 0, cy = 0, cz = 0, cm = 0;
                for (int j = 0; j < rt.x; j++) {
                    const mixed4 u = ld_img(sv, first + j);
                    const mixed m = u.w;                       // mass (0 for massless sites)
                    cx += u.x * m; cy += u.y * m; cz += u.z * m; cm += m;
                }
                const mixed w = rcp_(cm);
                cx *= w; cy *= w; cz *= w;
                scom[r] = mk4(cx, cy, cz, w);
                ke_com += ((double)cx * cx + (double)cy * cy + (double)cz * cz) * (double)cm;     // M v_com^2 (K :154)
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            const uint32_t m = meta[k];
            const uint32_t role = m & 3u, g = (m >> 2) & 255u;
            double cx = 0, cy = 0, cz = 0;
            if (use_com) { const mixed4 c = scom[m >> 21]; cx = c.x; cy = c.y; cz = c.z; }
            double val = 0.0;
            if (role == ROLE_NORMAL) {
                if (v[k].w != 0) {                               // K :161-168
                    const double rx = v[k].x - cx, ry = v[k].y - cy, rz = v[k].z - cz;
                    val = (rx * rx + ry * ry + rz * rz) * (double)mass[k];
                }
            } else if (role == ROLE_DRUDE) {                     // K :171-186 (one lane per pair)
                const int pl = k * TBLOCK + tid + (int)((m >> 10) & 2047u) - 1024;
                const mixed4 u = ld_img(sv, pl);
                const double r1x = v[k].x - cx, r1y = v[k].y - cy, r1z = v[k].z - cz;
                const double r2x = u.x - cx, r2y = u.y - cy, r2z = u.z - cz;
                const double mass1 = mass[k], mass2 = u.w;               // image .w = mass
                const double invTot = rcp_(mass1 + mass2);
                const double m1f = invTot * mass1, m2f = invTot * mass2;
                const double cmx = r1x * m1f + r2x * m2f, cmy = r1y * m1f + r2y * m2f, cmz = r1z * m1f + r2z * m2f;
                const double rlx = r2x - r1x, rly = r2y - r1y, rlz = r2z - r1z;
                val = (cmx * cmx + cmy * cmy + cmz * cmz) * (mass1 + mass2);
                ke_drude += (rlx * rlx + rly * rly + rlz * rlz) * (mass1 * mass2 * invTot);   // reduced mass = 1/invReducedMass (K :178, :185)
            }
            if constexpr (GB > 0) {
#pragma unroll
                for (int b = 0; b < GB; b++) ke_g[b] += (g == (uint32_t)b) ? val : 0.0;
            } else {
                // one pass per distinct group present in this wavefront (usually 1-3): butterfly-sum the lanes of
                // that group, lane 0 adds the sum to the wave's LDS bin.  The order depends on the data only.
                const bool has = role == ROLE_DRUDE || (role == ROLE_NORMAL && v[k].w != 0);
                unsigned long long rem = __ballot(has);
                while (rem) {
                    const int src = __ffsll((long long)rem) - 1;
                    const uint32_t g0 = __shfl(g, src, 64);
                    const bool mine = has && g == g0;
                    const double sg = wave_sum(mine ? val : 0.0);
                    if ((tid & 63) == 0) wbins[g0] += sg;
                    rem &= ~__ballot(mine);
                }
            }
        }
        lds_read = true;
    }
    if (lds_read) __syncthreads();       // LDS image is reused by the next tile
}


// MULTI: the in-kernel chain may have 2-4 links (chainN_run: ~100 registers of its own).  Its own instantiation, so that the
// one-link kernels keep their register count; and in it no wavefront issues its first tile's loads before the chain is done --
// the image of a tile in flight and the chain's links together would not fit three work-groups per compute unit.
template <int PREC, int OPS, int GB, bool MULTI = false>
__global__ __launch_bounds__(TBLOCK, TGNH_MINWAVES) void tile_kernel(const TileArgs a) {
    typedef typename Prec<PREC>::mixed mixed;
    typedef OpsOf<OPS> O;
    constexpr bool DO_SCALE = O::DO_SCALE, DO_KE = O::DO_KE, POS = O::POS;

    __shared__ double s_scale[MAX_GROUPS + 2];           // velocity scale factors of this launch (80 B: keeps smem 16-B aligned)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int G = a.num_groups;
    TileEnv<PREC, GB> e;
    e.init(a, smem, s_scale, POS && a.hardwall != 0);
    TilePattern tp;
    auto load_tile = [&](int tt, TileIn<PREC>& in) { tile_load<PREC, OPS, 0, OpsOf<OPS>::POS || OpsOf<OPS>::VEL_W>(a, a.reverse ? a.num_tiles - 1 - tt : tt, in, tp); };

    TileIn<PREC> cur;
    TRACE(0);
    if (a.commit_len > 0 && blockIdx.x == 0) {            // take over the thermostat block an in-kernel chain staged
        for (int i = tid; i < a.commit_len; i += TBLOCK)
            if (i < a.commit_skip || i >= a.commit_skip + a.commit_skip_n) a.commit_dst[i] = a.commit_src[i];
    }
    // ---- scale factors.  With a one-link chain the Nose-Hoover update itself runs here (A5): every work-group
    // computes the same factors from the same summed kinetic energies (fp64, deterministic); work-group 0 alone
    // writes the advanced thermostat block -- to a staging copy, because work-groups of this launch may start after
    // work-group 0 has finished.  One wavefront per work-group runs the chain, and runs it BEFORE issuing its own
    // tile loads: behind them, its wait for the thermostat state would be a wait for the whole tile (the counter
    // of outstanding loads completes in order), and every wavefront of the work-group would stand at the barrier
    // below for the latency of the memory phase PLUS the chain.  This way the chain (~3.5 us) hides behind the other
    // three wavefronts' loads.  (Wavefront 0 everywhere: the dispatcher starts consecutive work-groups of a compute
    // unit on consecutive SIMDs -- HW_ID, tools/trace_probe.py -- so the resident chains already sit on different
    // SIMDs; rotating the wavefront by residency slot made two of three collide.)
    const bool chain_wave = DO_SCALE && a.chain_on && tid < 64;
    const bool have_tile = (int)blockIdx.x < a.num_tiles;
    // sum_rows == 2 (many partial rows, no chain launch): ALL four wavefronts read a quarter of the rows each, in
    // batches of 16 loads issued ahead of the tile loads, so the row read costs one or two memory latencies that the
    // tile loads overlap -- read by the chain wavefront alone it was a chain of L2 misses on the critical path.
    // Flat view of the rows as in chain_sum_rows: lane l < W of wavefront w starts at element w W + l, stride 4 W.
    double racc = 0.0;
    int rcol = -1;
    __shared__ double s_part[TBLOCK / 64][CHAIN_INLINE_SUM_NT];
    if (DO_SCALE && a.chain_on && a.sum_rows == 2) {
        const int NT = G + 2, W = 64 - 64 % NT, lane = tid & 63;
        if (lane < W) {
            rcol = lane % NT;
            const int n = a.chain.nparts * NT, stride = (TBLOCK / 64) * W;
            for (int f0 = (tid >> 6) * W + lane; f0 < n; f0 += 16 * stride) {
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) { const int f = f0 + k * stride; v[k] = f < n ? a.chain.partials[f] : 0.0; }
#pragma unroll
                for (int k = 0; k < 16; k++) racc += v[k];
            }
            const double* big = a.chain.partials + (size_t)GRID_CAP * NT;
            for (int f = (tid >> 6) * W + lane; f < a.chain.nbig * NT; f += stride) racc += big[f];
        }
    }
    if (!MULTI && !chain_wave && have_tile) load_tile(blockIdx.x, cur);
    if (DO_SCALE && a.chain_on && a.sum_rows == 2) {
        const int NT = G + 2;
        for (int b = 0; b < NT; b++) {
            const double tb = wave_sum(rcol == b ? racc : 0.0);
            if ((tid & 63) == 0) s_part[tid >> 6][b] = tb;
        }
        __syncthreads();
    }
    TRACE(1);
    int trace_tile = 0; (void)trace_tile;
#ifdef TGNH_TRACE
    if ((threadIdx.x & 63) == 0) {          // slots 11/12: where the hardware put wavefronts 0 and 1 (HW_ID, XCC_ID)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (threadIdx.x == 0) g_trace[blockIdx.x * 16 + 11] = ((unsigned long long)xcc << 32) | hw;
        if (threadIdx.x == 64) g_trace[blockIdx.x * 16 + 12] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    if (DO_SCALE) {
        const int NT = G + 2;
        if (a.chain_on) {
            if (chain_wave) {
                const ChainLayout& L = a.chain.L;
                const bool write = blockIdx.x == 0;
                const int itg = tid & 63;
                TRACE(13);
                const bool one_link = !MULTI || L.C == 1;            // (2-4 links: chainN_run reads its state itself)
                Chain1Regs creg{};
                if (itg < NT) { if (one_link) creg = chain1_load(a.chain, a.st_in, itg); else creg.ke = a.st_in[chain_ke_src(a.chain) + itg]; }
                if (a.x_wait) {                                      // sharded: everybody's sums arrive by mailbox
                    const double tot = xchg_wait_sum(a.chain.x, NT, itg, reinterpret_cast<double*>(smem));   // the images are not in use yet
                    creg.ke = tot;
                    if (itg < NT) s_scale[itg] = tot;                // parked for the KESum below (same wavefront: in order)
                    if (write && itg < NT) a.chain.st[L.off_ke_red + itg] = tot;   // nobody reads it there in this launch
                }
                if (a.sum_rows) {                                    // no chain launch: the rows are summed in this launch
                    double mine = 0.0, ks = 0.0;
                    if (a.sum_rows == 2) {                           // the four wavefronts' quarters, in wavefront order
                        for (int b = 0; b < NT; b++) {
                            double tb = 0.0;
#pragma unroll
                            for (int w = 0; w < TBLOCK / 64; w++) tb += s_part[w][b];
                            mine = itg == b ? tb : mine;
                            ks += tb;
                        }
                    } else {
                        chain_sum_rows(a.chain, itg, &mine, &ks);      // a handful of rows: this wavefront alone
                    }
                    creg.ke = mine;
                    if (write && itg < NT) a.chain.st[L.off_ke_red + itg] = mine;   // nobody reads it there in this launch
                    if (write && itg == 63) a.st_out[L.off_kesum] = 0.5 * ks;       // Cu :493-497
                } else if (write && itg == 63) {                     // Cu :493-497
                    double s = 0.0;
                    for (int i = 0; i < NT; i++) s += a.x_wait ? s_scale[i] : a.st_in[chain_ke_src(a.chain) + i];
                    a.st_out[L.off_kesum] = 0.5 * s;
                }
                if (write && itg < NT && !a.chain.ke_carry && creg.ke != creg.ke) atomicOr(a.status, 16u);     // a NaN sum (a tail sum that gave up, here or on a peer rank): chain_prologue's check
                if (MULTI && !one_link) chainN_run(a.chain, a.st_in, a.st_out, write, s_scale, itg, creg.ke);
                else if (itg < NT) {
                    if (L.c1_quirk) chain1q_run(a.chain, creg, a.st_out, write, s_scale, itg);
                    else chain1_run(a.chain, creg, a.st_out, write, s_scale, itg);
                }
                TRACE(14);
                if (have_tile) load_tile(blockIdx.x, cur);
            } else if (MULTI && have_tile) load_tile(blockIdx.x, cur);
        } else {
            if (tid < NT) s_scale[tid] = a.scale[tid];
            if (MULTI && have_tile) load_tile(blockIdx.x, cur);
        }
        __syncthreads();
        e.s_com = (mixed)s_scale[G]; e.s_drude = (mixed)s_scale[G + 1];
    }
    TRACE(2);
    for (int t = blockIdx.x; t < a.num_tiles; t += gridDim.x) {
        const bool more = t + (int)gridDim.x < a.num_tiles;
        tile_body<PREC, OPS, GB>(a, e, cur, trace_tile);
        TRACE(6 + 4 * trace_tile);
#ifdef TGNH_TRACE
        trace_tile++;
#endif
        if (more) load_tile(t + gridDim.x, cur);
    }
    if (DO_KE) ke_reduce<PREC, GB, false>(a, e);
    TRACE(15);
}

// ---------------------------------------------------------------------------
// wke_kernel: the kinetic-energy passes (KE; kick+KE; kick+KE without a velocity store) over WAVE tiles.
//
// What such a pass needs per slot beyond its own velocity is its Drude partner (a lane or so away) and its molecule's
// centre-of-mass velocity.  tile_kernel gets both from an LDS image of a 512-slot tile shared by four wavefronts -- store,
// barrier, one thread per molecule walks its slots (12 of 64 lanes busy, five dependent LDS reads each), barrier, look-ups,
// barrier -- and the work-group issues its next loads only then.  Here a wavefront owns <= 64 consecutive slots that never
// cut a molecule or a pair (tgnh_internal.h) and a PRIVATE 2 KiB LDS image: it stores its velocities and masses (component
// arrays, conflict-free) and every lane sums its OWN molecule from the image, in slot order -- the arithmetic and the order of
// tile_kernel's walk, with all lanes busy and no dependence between wavefronts.  A wavefront's LDS operations are processed
// in order, so nothing waits for a barrier; the next tile's global loads are in flight while this one is worked on, the tile
// bounds in scalar registers two tiles ahead.
// Reference: K :82-113 (COM), :119-133 (relative velocities), :138-200 (bins), :307-365 (the kick); Ref :439-460.
// ---------------------------------------------------------------------------
template <int PREC> struct WaveIn {
    typename Prec<PREC>::mixed4 v;
    uint32_t meta;
    long long fx, fy, fz;
};


template <int PREC, int OPS>
__device__ __forceinline__ void wave_load(const TileArgs& a, const int ws, const int n, const bool patterned, const uint32_t pword, const int lane, WaveIn<PREC>& in) {
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::mixed4 mixed4;
    const mixed4* __restrict__ velm = reinterpret_cast<const mixed4*>(a.velm);
    const int idx = ws + lane;
    mixed4 v = mk4((mixed)0, (mixed)0, (mixed)0, (mixed)0);      // a padding lane: massless, role normal, a molecule of its own -- contributes nothing
    uint32_t meta = 64u << 10;
    long long fx = 0, fy = 0, fz = 0;
    if (lane < n) {
        v = velm[idx];
        if (!patterned) meta = a.wmeta[idx];
        if (OPS & OP_KICK) {
            fx = a.force[idx];
            fy = a.force[idx + a.padded];
            fz = a.force[idx + 2 * a.padded];
        }
    }
    if (patterned && lane < n) meta = pword;
    in.v = v; in.meta = meta; in.fx = fx; in.fy = fy; in.fz = fz;
}

template <int GB, bool LEAN, int NTH>
__device__ __forceinline__ bool collect_rows(const TileArgs& a, const int tid, const int grid, const int NT, const unsigned long long want,
                                             double (&acc)[GB + 2]);

template <int PREC, int OPS, int GB>
__global__ __launch_bounds__(TBLOCK, TGNH_MINWAVES) void wke_kernel(const TileArgs a) {
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::mixed4 mixed4;
    constexpr bool DO_KICK = (OPS & OP_KICK) != 0, STORE = DO_KICK && !(OPS & OP_NOSTORE);
    static_assert(GB > 0, "register bins only");
    __shared__ double sred[TBLOCK / 64][GB + 2];
    __shared__ mixed s_img[TBLOCK / 64][4][WAVE_SLOTS];          // per wavefront: x[], y[], z[], mass[]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    mixed* const ix = s_img[wv][0]; mixed* const iy = s_img[wv][1]; mixed* const iz = s_img[wv][2]; mixed* const im = s_img[wv][3];
    const int G = a.num_groups;
    const bool use_com = a.use_com != 0;
    const mixed fscale = (mixed)(0.5 * a.dt / 4294967296.0);     // Cu :295
    mixed4* __restrict__ velm = reinterpret_cast<mixed4*>(a.velm);
    if (a.commit_len > 0 && blockIdx.x == 0) {            // take over the thermostat block an in-kernel chain staged (as tile_kernel)
        for (int i = tid; i < a.commit_len; i += TBLOCK)
            if (i < a.commit_skip || i >= a.commit_skip + a.commit_skip_n) a.commit_dst[i] = a.commit_src[i];
    }
    double ke_g[GB], ke_com = 0.0, ke_drude = 0.0;
#pragma unroll
    for (int b = 0; b < GB; b++) ke_g[b] = 0.0;
    // tail sum: this launch's number (the tag of its rows), read before anything is handed in
    const unsigned gen0 = a.tail_sum ? __hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;

    // Wave tile of wavefront wv in round r: (r gridDim.x + blockIdx.x) 4 + wv -- a work-group streams 4 consecutive wave
    // tiles.  Everything about WHICH tile is wavefront-uniform (scalar registers, scalar loads): the bounds of the tile after
    // next are fetched while this one is worked on, so the vector loads of the next tile never wait for an index.
    const int nw = a.num_wtiles, stride = (int)gridDim.x * (TBLOCK / 64);
    struct Bounds { int ws, y, n; };        // y = the tile's largest molecule | pattern word << 8 (wave_word)
    auto bounds = [&](const int ww, Bounds& b) {
        const int2* t = a.wave_tile + (a.reverse ? nw - 1 - ww : ww);
        b.ws = t[0].x; b.y = t[0].y; b.n = t[1].x - b.ws;
    };
    auto work = [&](const WaveIn<PREC>& cur, const Bounds& bd) {
        mixed4 v = cur.v;
        const uint32_t m = cur.meta;
        const mixed mass = v.w != 0 ? rcp_(v.w) : (mixed)0;
        if (DO_KICK) {                                                   // A7, per particle (tile_body); w = 0: c = 0, v unchanged
            const mixed c = fscale * v.w;
            v.x += c * force_as(cur.fx, (mixed)0);
            v.y += c * force_as(cur.fy, (mixed)0);
            v.z += c * force_as(cur.fz, (mixed)0);
        }
        if (STORE && lane < bd.n) velm[bd.ws + lane] = v;
        const uint32_t role = m & 3u, g = (m >> 2) & 255u;
        // the wavefront's image (its own LDS operations are processed in order: the reads below see these stores, and the
        // stores of the next tile come after this tile's reads -- the fences only keep the compiler from reordering them)
        ix[lane] = v.x; iy[lane] = v.y; iz[lane] = v.z; im[lane] = mass;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- molecular centre-of-mass velocity (K :86-111): every lane sums its own molecule, in slot order
        mixed cx = 0, cy = 0, cz = 0;
        if (use_com) {
            const int j = (int)((m >> 17) & 63u), n1 = (int)((m >> 23) & 63u);
            const int first = lane - j;
            mixed px = 0, py = 0, pz = 0, pm = 0;
            for (int k = 0; k < (bd.y & 255); k++) {                          // (bd.maxn: the tile's largest molecule, a scalar)
                if (k <= n1) {
                    const mixed um = im[first + k];
                    px += ix[first + k] * um; py += iy[first + k] * um; pz += iz[first + k] * um; pm += um;
                }
            }
            const mixed wq = rcp_(pm);                                   // (a padding lane: pm = 0, unused)
            cx = px * wq; cy = py * wq; cz = pz * wq;
            if (j == 0 && lane < bd.n)                                   // once per molecule: M v_com^2 (K :154)
                ke_com += ((double)cx * cx + (double)cy * cy + (double)cz * cz) * (double)pm;
        }
        // ---- bins (K :138-200 ; Ref :439-460).  Every massive slot adds m |v - v_com|^2 to its group's bin; a pair's
        // two terms together are (m1 + m2) |cm - v_com|^2 + mu |v2 - v1|^2 (K :171-186 splits them that way), so the Drude
        // lane moves the second part, mu |v2 - v1|^2, from the group's bin to the Drude bin: the partner is needed for that
        // difference only
        const double rx = v.x - cx, ry = v.y - cy, rz = v.z - cz;          // (in the velocities' own precision, as tile_body)
        double val = v.w != 0 ? (rx * rx + ry * ry + rz * rz) * (double)mass : 0.0;
        if (role == ROLE_DRUDE) {                                        // one lane per pair
            const int pl = lane + (int)((m >> 10) & 127u) - 64;
            const double dx = ix[pl] - v.x, dy = iy[pl] - v.y, dz = iz[pl] - v.z;
            const double mass1 = mass, mass2 = im[pl];
            const double mu = mass1 * mass2 * rcp_(mass1 + mass2);       // reduced mass = 1/invReducedMass (K :178, :185)
            const double d = (dx * dx + dy * dy + dz * dz) * mu;
            ke_drude += d;
            val -= d;
        }
#pragma unroll
        for (int b = 0; b < GB; b++) ke_g[b] += (g == (uint32_t)b) ? val : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int w = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (TBLOCK / 64) + wv);
    Bounds b0{}, b1{}, b2{};
    WaveIn<PREC> A, B;
    PatternWord pw;
    auto load = [&](const Bounds& b, WaveIn<PREC>& in) {
        const bool patterned = pw.of(a, (uint32_t)b.y >> 8, lane);
        wave_load<PREC, OPS>(a, b.ws, b.n, patterned, pw.word, lane, in);
    };
    if (w < nw) { bounds(w, b0); load(b0, A); }
    if (w + stride < nw) bounds(w + stride, b1);
    while (w < nw) {                                                     // two tiles per trip: the register images alternate
        if (w + 2 * stride < nw) bounds(w + 2 * stride, b2);
        if (w + stride < nw) load(b1, B);                                // in flight while A is worked on
        work(A, b0);
        w += stride;
        if (w >= nw) break;
        if (w + 2 * stride < nw) bounds(w + 2 * stride, b0);
        if (w + stride < nw) load(b2, A);
        work(B, b1);
        w += stride;
        b1 = b0; b0 = b2;                                                // (scalar moves)
    }
    // ---- one row of partial sums per work-group: 64-lane sums, one LDS hop, fixed order (ke_reduce's layout)
#pragma unroll
    for (int b = 0; b < GB; b++) ke_g[b] = wave_sum(ke_g[b]);
    ke_com = wave_sum(ke_com);
    ke_drude = wave_sum(ke_drude);
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < GB; b++) sred[wv][b] = ke_g[b];
        sred[wv][GB] = ke_com;
        sred[wv][GB + 1] = ke_drude;
    }
    __syncthreads();
    if (tid < GB + 2) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < TBLOCK / 64; k++) t += sred[k][tid];
        const int b = tid < GB ? tid : G + (tid - GB);                 // thermostat of this thread's sum
        if (tid >= GB || tid < G) {
            if (a.tail_sum) {                                            // a tagged cell: data and "it is there" in one 8-byte store, two per double
                unsigned long long* cell = a.rows + row_word((int)blockIdx.x, 2 * b);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(t), tg = (unsigned long long)(gen0 + 1u) << 32;
                __hip_atomic_store(cell, tg | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(cell + 64, tg | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else a.partials[(size_t)blockIdx.x * (G + 2) + b] = t;
        }
    }
    // ---- tail sum: work-group 0 -- which therefore ends last -- collects every work-group's row in row order (fixed order:
    // reproducible bits) and leaves the sums where the row-sum launch would have.  Only it waits, so the grid need not be
    // resident at once; the polling is bounded (status bit 3, as step_kernel's).
    if (a.tail_sum && blockIdx.x == 0) {
        __shared__ double s_tail[TBLOCK / 64][GB + 2];
        const int NT = G + 2;
        double acc[GB + 2];
#pragma unroll
        for (int b = 0; b < GB + 2; b++) acc[b] = 0.0;
        const bool ok = collect_rows<GB, false, TBLOCK>(a, tid, (int)gridDim.x, NT, (unsigned long long)(gen0 + 1u), acc);
        if (!ok) atomicOr(a.status, 16u);                                // a row never came (bounded polling): status bit 4, the host's failure
#pragma unroll
        for (int b = 0; b < GB + 2; b++) {
            if (b < NT) {
                const double t = wave_sum(acc[b]);
                if (lane == 0) s_tail[wv][b] = t;
            }
        }
        // (the barrier of the hand-over doubles as the vote: incomplete sums are not left where the all-reduce and the chain
        // would take them for kinetic energies -- NaN instead, so that nothing integrates on with a partial sum during the
        // up to 64 steps until the host reads the status word; step_meet withholds its send in the same situation)
        const bool all_ok = __syncthreads_and(ok ? 1 : 0) != 0;
        if (tid < NT) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < TBLOCK / 64; k++) t += s_tail[k][tid];
            a.ke_red[tid] = all_ok ? t : __longlong_as_double(0x7ff8000000000000ll);
        }
        if (tid == 0) a.sync[1] = gen0 + 1u;                             // the next launch's rows carry the next tag
    }
}



// ---------------------------------------------------------------------------
// step_kernel: a whole time step of the deferred pass structure in ONE launch (TGNH_FLAG_RESIDENT_STEP).
//
//   pass 1   half kick (unstored) + kinetic-energy sums over the work-group's tiles          (Cu :384-388, :474-488)
//   meet     every work-group leaves its row of sums as tagged cells (data and "it is there" in one 8-byte store);
//            work-group 0 collects the rows (fixed order: reproducible bits), and sends the sums to the mailbox of
//            every rank -- its own included; unsharded, the handle's private one-rank mailbox -- where every
//            work-group of every rank waits for all ranks' sums.  No read-modify-write atomics anywhere: 768
//            work-groups arriving at one counter cost ~70 us, plain tagged stores and polling loads a few
//   chain    both thermostat half steps back to back, by one wavefront of every work-group      (Cu :433-652 twice)
//   pass 2   the kick again, rescale, half kick, drift, hard wall over the same tiles           (Cu :351-376)
//
// The grid is the work-groups that are resident at once (occupancy x CUs), so work-group 0's wait cannot deadlock as
// long as this launch has its share of the device to itself; every wait is bounded all the same (status bits 2 / 3,
// never a hung device).  Pass 2 walks the work-group's tiles backwards: its first tile is pass 1's last, whose
// velocities, forces and index words are still in registers -- only its positions are fetched, and that before the
// meeting, which hides them.  At shard sizes (<= 2 tiles per work-group) most of the step's state therefore never
// leaves the chip between the passes.  The thermostat block is advanced in place by work-group 0: every work-group
// reads it before it hands in its row, and work-group 0 writes only after it has everybody's.
// ---------------------------------------------------------------------------
// The two passes are template parameters, so the same kernel also runs the thermostat halves of the reference's own pass
// structure (velocities never lag: what the OpenMM glue may use) as one launch each:
//   STEP_DEFER        kick+KE (unstored)  |  kick again, rescale, kick, drift     both chain halves   a whole deferred step
//   STEP_PLAIN_BEGIN  KE                  |  rescale, kick, drift                 one half            Cu :336-376
//   STEP_PLAIN_END    kick+KE (unstored)  |  kick again, rescale                  one half            Cu :384-402
//   STEP_SPLIT_BEGIN  KE                  |  rescale, kick, posDelta              one half            Cu :336-360 (constraints)
//   STEP_SPLIT_END    KE                  |  rescale                              one half            Cu :394-402 (constraints)
enum : int { STEP_DEFER = 0, STEP_PLAIN_BEGIN = 1, STEP_PLAIN_END = 2, STEP_SPLIT_BEGIN = 3, STEP_SPLIT_END = 4, STEP_KINDS = 5 };
constexpr int step_ops1(int kind) {
    return (kind == STEP_DEFER || kind == STEP_PLAIN_END) ? (OP_KICK | OP_KE | OP_NOSTORE) : OP_KE;
}
constexpr int step_ops2(int kind) {
    return kind == STEP_DEFER ? (OP_PREKICK | OP_SCALE | OP_KICK | OP_DRIFT)
         : kind == STEP_PLAIN_BEGIN ? (OP_SCALE | OP_KICK | OP_DRIFT)
         : kind == STEP_PLAIN_END ? (OP_PREKICK | OP_SCALE)
         : kind == STEP_SPLIT_BEGIN ? (OP_SCALE | OP_KICK | OP_POSDELTA)
         : OP_SCALE;
}

// (Measured and dropped, profiles/r02_resident_tuning.md: a second register image to load a work-group's next tile under
// the current one -- in both passes: 198 VGPRs, occupancy 2; in pass 1 alone: free in registers, no gain -- and tiles cut
// to N / (k x work-groups) slots for equal walks.  A pass costs ~2 us of a compute unit's time per tile whether a
// work-group walks one tile or two: it is the unit's three resident work-groups that overlap each other, not a
// work-group its own tiles.  Pass 2 already moves its 73 MB at the 6.6 TB/s the Infinity Cache gives.)
template <int PREC, int GB, int KIND>
__global__ __launch_bounds__(TBLOCK, TGNH_MINWAVES) void step_kernel(const TileArgs a) {
    typedef typename Prec<PREC>::mixed mixed;
    constexpr int STEP_OPS1 = step_ops1(KIND), STEP_OPS2 = step_ops2(KIND);
    __shared__ double s_scale[MAX_GROUPS + 2];
    __shared__ double s_part[TBLOCK / 64][CHAIN_INLINE_SUM_NT];
    __shared__ double s_x[64 + XCHG_MAX_WORLD * CHAIN_INLINE_SUM_NT];      // scratch of the sums and the exchange: the images stay intact
    __shared__ int s_go;
    __shared__ unsigned s_gen;
    __shared__ unsigned long long s_seq1;                  // the number of the exchange this launch sends and waits for
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, G = a.num_groups, NT = G + 2;
    const int grid = (int)gridDim.x;
    const bool chain_wave = tid < 64;
    const int itg = tid & 63;
    if (a.census) {
        // One-time check at tgnh_create that a grid of this size really is resident all at once (the occupancy API can be
        // one work-group per compute unit high, MI355X_MICROARCH.md "Correctness boundaries"): every work-group checks in
        // at a counter and waits, bounded, until all have; one that gives up says so.  Nothing else is touched.
        if (tid == 0) {
            __hip_atomic_fetch_add(&a.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned n = 0;
            while (__hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++n < CENSUS_SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(16);
            if (__hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) atomicOr(&a.sync[3], 1u);
        }
        return;
    }
    TileEnv<PREC, GB> e;
    e.init(a, smem, s_scale, OpsOf<STEP_OPS2>::POS && a.hardwall != 0);
    auto tile_of = [&](int tt) { return a.reverse ? a.num_tiles - 1 - tt : tt; };

    // this launch's number (the tag of its rows), the exchange it will wait for and this wavefront's thermostat state: read
    // before anything is handed in.  The thermostat block is advanced IN PLACE by work-group 0 once it holds every row, so
    // every work-group must have READ the block before its row goes out: the loads are issued here, ahead of the first
    // tile's (loads return in order), and their registers are pinned just before ke_reduce's tagged stores below, which the
    // same wavefront issues -- the order is program order plus a data dependency, not a matter of latencies.
    unsigned gen0 = 0;
    unsigned long long seq0 = 0;
    Chain1Regs creg{};
    if (chain_wave) {
        gen0 = __hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seq0 = __hip_atomic_load(a.chain.x.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (itg < NT) creg = chain1_load(a.chain, a.st_in, itg);
    }

    // ---- pass 1
    TRACE(0);
    TileIn<PREC> cur;
    int tt = blockIdx.x;                                   // grid <= num_tiles: every work-group has a tile
    TilePattern tp;
    tile_load<PREC, STEP_OPS1>(a, tile_of(tt), cur, tp);
    for (;;) {
        tile_body<PREC, STEP_OPS1, GB>(a, e, cur, 4);      // (trace slots >= 16: not recorded)
        if (tt + grid >= a.num_tiles) break;
        tt += grid;
        tile_load<PREC, STEP_OPS1>(a, tile_of(tt), cur, tp);
    }
    const int tt_last = tt;                                // stays in `cur`; its velocity image and COM table stay in LDS
    TRACE(1);
    MeetShared sh{s_scale, s_part, s_x, &s_go, &s_gen, &s_seq1, nullptr};
    if (!step_meet<PREC, GB>(a, e, gen0, seq0, creg, sh, [&] { tile_load<PREC, STEP_OPS2, STEP_OPS1>(a, tile_of(tt_last), cur, tp); }))
        return;                                            // an exchange timed out: reported by the status word; nothing is stored
    e.s_com = (mixed)s_scale[G]; e.s_drude = (mixed)s_scale[G + 1];
    TRACE(9);

    // ---- pass 2, backwards from the held tile
    tt = tt_last;
    for (bool held = true;; held = false) {                // (one call site: two cost 40 VGPRs and a work-group per CU)
        tile_body<PREC, STEP_OPS2, GB>(a, e, cur, 0, held);  // the held tile: image and COM table of pass 1
        if (tt - grid < 0) break;
        tt -= grid;
        tile_load<PREC, STEP_OPS2>(a, tile_of(tt), cur, tp);
    }
    TRACE(15);
}

// ---------------------------------------------------------------------------
// wstep_kernel: step_kernel's whole deferred time step (STEP_DEFER) over WAVE tiles -- wke_kernel's structure for both passes.
//
// A wavefront owns <= 64 consecutive slots and a private LDS image; nothing in a pass waits for another wavefront.  What that
// buys at shard sizes: a pass without barriers, and for a held tile a second pass that starts from registers.  The one-link
// instantiation takes 110 (single) / 121 (mixed, double) VGPRs = 4 wavefronts per SIMD = two 512-thread work-groups per compute
// unit: 512 work-groups = 262 144 slots are resident at once (tests/test_kernel_resources.py asserts the occupancy) -- at 625 k
// slots a wavefront walks 2.4 tiles forward and back, at 5 M slots 19 -- and for the tile it holds across the meeting: its kicked
// velocities, forces, index word, mass and centre-of-mass velocity are pass 1's, its partner's velocity is still in the
// wavefront's image, its positions were fetched before the meeting.  Same meeting (step_meet), same arithmetic per slot as
// tile_body / wke_kernel, same fixed order of every sum.  Topologies without wave tiles (a molecule longer than a wavefront,
// more than 8 temperature groups) and the other step kinds run step_kernel.
//   pass 1   half kick (unstored) + kinetic-energy sums            (Cu :384-388, :474-488)
//   meet     rows -> work-group 0 -> mailboxes -> both chain halves (Cu :433-652 twice)
//   pass 2   the kick again, rescale, half kick, drift, hard wall   (Cu :351-376 ; K :249-301, :307-365, :435-466, :471-574)
// ---------------------------------------------------------------------------

template <int PREC, int GB, bool MULTI = false>
__global__ __launch_bounds__(WBLOCK) void wstep_kernel(const TileArgs a) {
    typedef typename Prec<PREC>::mixed mixed;
    static_assert(GB > 0, "register bins only");
    __shared__ double s_scale[MAX_GROUPS + 2];
    __shared__ double s_part[WBLOCK / 64][CHAIN_INLINE_SUM_NT];
    __shared__ double s_x[64 + XCHG_MAX_WORLD * CHAIN_INLINE_SUM_NT];
    __shared__ int s_go;
    __shared__ unsigned s_gen;
    __shared__ unsigned long long s_seq1;
    __shared__ mixed s_img[WBLOCK / 64][7][WAVE_SLOTS];          // per wavefront: velocity x, y, z, mass; position x, y, z (hard wall)
    __shared__ double s_block[256];                              // chains of 2-4 links: the thermostat block as it was at entry
    const int tid = threadIdx.x, lane = tid & 63, G = a.num_groups, NT = G + 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool chain_wave = tid < 64;
    const int itg = tid & 63;
    if (a.census) {                                        // residency check at tgnh_create, as step_kernel's
        if (tid == 0) {
            __hip_atomic_fetch_add(&a.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned n = 0;
            while (__hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++n < CENSUS_SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(16);
            if (__hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) atomicOr(&a.sync[3], 1u);
        }
        return;
    }
    TileEnv<PREC, GB> e;                                   // the kinetic-energy bins, in the shape ke_reduce takes them
    e.tid = tid; e.G = G; e.smem = nullptr; e.wbins0 = nullptr; e.s_scale = s_scale;
    e.clear_ke();
    WaveStep<PREC, GB> ws(a, &e, s_scale, &s_img[wv][0][0], lane);    // the per-tile work (tgnh_tile_device.h)
    typedef WaveBounds Bounds;
    auto bounds = [&](const int ww, Bounds& b) { ws.bounds(ww, b); };
    auto load_vf = [&](const Bounds& b, WStepIn<PREC>& in) { ws.load_vf(b, in); };
    auto load_x = [&](const Bounds& b, WStepIn<PREC>& in) { ws.load_x(b, in); };
    auto prepare = [&](WStepIn<PREC>& t, const Bounds& bd, const bool ke) { ws.template prepare<true>(t, bd, ke); };
    auto finish = [&](WStepIn<PREC>& t, const Bounds& bd) { ws.finish(t, bd); };
    auto wfence = [] { WaveStep<PREC, GB>::wfence(); };

    // launch number, exchange number and the thermostat state: read before anything is handed in (step_kernel)
    unsigned gen0 = 0;
    unsigned long long seq0 = 0;
    Chain1Regs creg{};
    if (chain_wave) {
        gen0 = __hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seq0 = __hip_atomic_load(a.chain.x.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (itg < NT && !(MULTI && a.chain.L.C > 1)) creg = chain1_load(a.chain, a.st_in, itg);
    }
    // Chains of 2-4 links: the whole block (<= 256 doubles, checked on the host) goes to LDS now -- work-group 0 advances it in
    // place once it holds every row, and a row leaves only behind the barrier in ke_reduce, which every thread reaches after
    // its load has landed and been stored here.
    if (MULTI && a.chain.L.C > 1 && tid < a.chain.L.total) s_block[tid] = a.st_in[tid];

    const int nw = a.num_wtiles, stride = (int)gridDim.x * (WBLOCK / 64);

    // ---- pass 1: this wavefront's tiles w0, w0 + stride, ...; the next tile's loads in flight while one is worked on
    TRACE(0);
    const int w0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (WBLOCK / 64) + wv);
    const bool have = w0 < nw;                             // (a wavefront beyond the last tile only takes part in the meeting)
    int w = w0;
    // One register image: no tile is loaded ahead of the one being worked on.  What hides a tile's load latency is the other
    // wavefronts of the SIMD -- four of them at this register count (a second image was measured at the same occupancy and
    // bought nothing; it would cost the fourth wavefront today).
    // Tile bounds are scalar loads one tile ahead of their use.
    Bounds b0{}, b1{};
    WStepIn<PREC> cur;
    if (have) {
        bounds(w, b0);
        if (w + stride < nw) bounds(w + stride, b1);
        load_vf(b0, cur);
    }
    while (have) {
#ifdef TGNH_TRACE
        if (w == w0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TRACE(3); }
#endif
        prepare(cur, b0, true);
#ifdef TGNH_TRACE
        if (w == w0) TRACE(4);
#endif
        if (w + stride >= nw) break;
        wfence();                                          // (this tile's image reads are done before the next tile's image is stored)
        w += stride;
        b0 = b1;
        if (w + stride < nw) bounds(w + stride, b1);
        load_vf(b0, cur);
    }
    const int w_last = w;                                  // stays in `cur` (kicked velocities, mass, v_com), its image in LDS
    TRACE(1);
    MeetShared sh{s_scale, s_part, s_x, &s_go, &s_gen, &s_seq1, s_block};
    if (!step_meet<PREC, GB, true, WBLOCK, MULTI>(a, e, gen0, seq0, creg, sh, [&] {
            if (have) {
                load_x(b0, cur);
                if (w_last - stride >= 0) bounds(w_last - stride, b1);        // the way back: known long before it is needed
            }
        })) return;

    // ---- pass 2, backwards from the held tile
    TRACE(9);
    if (have) {
        w = w_last;
        finish(cur, b0);                                   // the held tile: everything but its positions is pass 1's
        TRACE(5);
        while (w - stride >= 0) {
            w -= stride;
            b0 = b1;
            if (w - stride >= 0) bounds(w - stride, b1);
            load_vf(b0, cur); load_x(b0, cur);
            prepare(cur, b0, false);
            finish(cur, b0);
        }
    }
    TRACE(15);
}

// ---------------------------------------------------------------------------
// chain_kernel: cross-work-group KE sum (fixed order) + Nose-Hoover chain (A5)
// One work-group.  TGNH: lane itg owns thermostat itg (Cu :558-650).
// dualNH: lane 0 runs the reference's coupled, interleaved arrays (Ref :467-504),
// including its indexing quirk when useDrudeNHChains is false.
// The chain variables are copied into registers (numNHChains <= 4, fully unrolled) or
// LDS (longer chains) for the S-fold loop and written back once: with them left in
// global memory every `etaDot[i] *= expfac` was a dependent HBM round trip.
// ---------------------------------------------------------------------------
constexpr int CHAIN_LDS_DOUBLES = 2048;

// The part of chain_kernel before the chain itself: commit of a staged block, fixed-order sum of the partial rows, the
// exchange's send / wait.  Shared with rowsum_kernel below.
__device__ __forceinline__ void chain_prologue(const ChainArgs& a, double (*sred)[MAX_GROUPS + 2], double* s_chain, double* s_ke) {
    const ChainLayout& L = a.L;
    const int NT = L.NT, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double* st = a.st;
    if (a.commit) {                  // take over the block an in-kernel chain staged (everything but the KE sums)
        for (int i = tid; i < L.total; i += BLOCK)
            if (i < L.off_ke_red || i >= L.off_ke_red + NT) st[i] = a.stage[i];
        __syncthreads();
    }
    if (a.do_sum) {
        // Fixed-order sum of the work-group partials: lane `tid` owns partials tid, tid+256, ...; every load of a
        // lane is issued before the first add (one memory latency, not one per partial), then a 64-lane sum
        // and a 4-wave LDS hop.  The order never depends on timing, so the sums are reproducible bit for bit.
        constexpr int PER = GRID_CAP / BLOCK;
        double acc[MAX_GROUPS + 2];
#pragma unroll
        for (int b = 0; b < MAX_GROUPS + 2; b++) acc[b] = 0.0;
        if (NT <= 4) {
            double val[PER][4];
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const int p = tid + j * BLOCK;
#pragma unroll
                for (int b = 0; b < 4; b++) val[j][b] = (p < a.nparts && b < NT) ? a.partials[(size_t)p * NT + b] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < PER; j++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[b] += val[j][b];
        } else {
#pragma unroll
            for (int b = 0; b < MAX_GROUPS + 2; b++) {
                if (b < NT) {
                    double val[PER];
#pragma unroll
                    for (int j = 0; j < PER; j++) { const int p = tid + j * BLOCK; val[j] = p < a.nparts ? a.partials[(size_t)p * NT + b] : 0.0; }
#pragma unroll
                    for (int j = 0; j < PER; j++) acc[b] += val[j];
                }
            }
        }
        for (int p = tid; p < a.nbig; p += BLOCK) {                  // rows of the molecules longer than a tile
#pragma unroll
            for (int b = 0; b < MAX_GROUPS + 2; b++)
                if (b < NT) acc[b] += a.partials[((size_t)GRID_CAP + p) * NT + b];
        }
#pragma unroll
        for (int b = 0; b < MAX_GROUPS + 2; b++) {
            if (b < NT) {
                const double s = wave_sum(acc[b]);
                if (lane == 0) sred[wv][b] = s;
            }
        }
        __syncthreads();
        if (tid < NT) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; w++) s += sred[w][tid];
            st[L.off_ke_red + tid] = s;
            s_ke[tid] = s;
        }
        __syncthreads();
    } else if (tid < NT) {
        s_ke[tid] = st[chain_ke_src(a) + tid];   // summed (and all-reduced) by an earlier launch, or carried over from the last chain
    }
    // A tail sum that gave up on a row leaves NaN (wke_kernel) and sets status bit 4 on ITS rank; behind an all-reduce every rank
    // holds that NaN now, and says so itself -- no rank integrates on silently with a clean status word
    // (not for sums carried over from the last chain, ke_carry: a system without any Drude pair carries the reference's own 0/0
    // in its Drude thermostat, harmless there -- Cu :597-605 with drudeDof = 0)
    if (tid < NT && a.status && !a.ke_carry && s_ke[tid] != s_ke[tid]) atomicOr(a.status, 16u);
    if (a.x_send) xchg_send(a.x, NT, tid, BLOCK, s_chain, tid < NT ? s_ke[tid] : 0.0);
    if (a.x_wait) {
        __syncthreads();
        if (tid < 64) {
            const double tot = xchg_wait_sum(a.x, NT, tid, s_chain);
            if (tid < NT) { s_ke[tid] = tot; st[L.off_ke_red + tid] = tot; }
        }
        __syncthreads();
    }
}

// The row sum alone (do_chain == 0: the chain itself runs inside the next rescale launch): its own small kernel.  A launch
// starts with a cold instruction cache, and behind a pass that has streamed a gigabyte through the L2 and the Infinity Cache
// its code comes from HBM: what a one-work-group launch costs is mostly the cache lines of code on its path.  Inside
// chain_kernel (27 000 lines of ISA with every chain length inlined) that path jumped across the whole kernel.
__global__ __launch_bounds__(BLOCK) void rowsum_kernel(const ChainArgs a) {
    __shared__ double sred[BLOCK / 64][MAX_GROUPS + 2];
    __shared__ double s_chain[2 * XCHG_MAX_WORLD * XCHG_NT_PAD > 64 ? 2 * XCHG_MAX_WORLD * XCHG_NT_PAD : 64];
    __shared__ double s_ke[MAX_GROUPS + 2];
    chain_prologue(a, sred, s_chain, s_ke);
}

__global__ __launch_bounds__(BLOCK) void chain_kernel(const ChainArgs a) {
    __shared__ double sred[BLOCK / 64][MAX_GROUPS + 2];
    __shared__ double s_chain[CHAIN_LDS_DOUBLES];
    __shared__ double s_ke[MAX_GROUPS + 2];
    const ChainLayout& L = a.L;
    const int NT = L.NT, tid = threadIdx.x;
    double* st = a.st;
    CHAIN_TRACE(0);
    chain_prologue(a, sred, s_chain, s_ke);
    if (!a.do_chain) return;
    if (!a.do_sum) __syncthreads();
    CHAIN_TRACE(1);
    if (L.mode == TGNH_MODE_TGNH && L.C > 4 && L.C <= 16 && a.lanes) {
        chain_lanes_run(a, st, tid, BLOCK, s_ke);                    // 5-16 links: a link per lane, in registers
        if (tid == 0) {                                              // Cu :493-497
            double s = 0.0;
            for (int i = 0; i < NT; i++) s += s_ke[i];
            st[L.off_kesum] = 0.5 * s;
        }
    } else if (L.mode == TGNH_MODE_TGNH) {
        // real thermostats on lanes 0..NT-2 of wave 0, the Drude thermostat on lane 0 of wave 1: the two code
        // paths then run side by side on two SIMDs instead of one after the other under one exec mask
        int itg = -1;
        if (tid < NT - 1) itg = tid;
        else if (tid == 64) itg = NT - 1;
        if (itg >= 0) {
            switch (L.C) {
                case 1: { Chain1Regs r = chain1_load(a, a.st, itg); r.ke = s_ke[itg]; chain1_run(a, r, a.st, true, nullptr, itg); } break;   // the arithmetic of the in-kernel chain
                case 2: run_tgnh<2>(a, a.st, a.st, true, nullptr, itg, s_chain, s_ke[itg]); break;
                case 3: run_tgnh<3>(a, a.st, a.st, true, nullptr, itg, s_chain, s_ke[itg]); break;
                case 4: run_tgnh<4>(a, a.st, a.st, true, nullptr, itg, s_chain, s_ke[itg]); break;
                default: run_tgnh<0>(a, a.st, a.st, true, nullptr, itg, s_chain, s_ke[itg]); break;   // host checked NT*(4C+1) <= CHAIN_LDS_DOUBLES
            }
        }
        if (tid == 0) {                                              // Cu :493-497
            double s = 0.0;
            for (int i = 0; i < NT; i++) s += s_ke[i];
            st[L.off_kesum] = 0.5 * s;
        }
    } else if (L.C == 1) {
        // one link: with useDrudeNHChains two independent one-link chains, the code of the TGNH ones (Chain1Map);
        // without, the same two lanes coupled by one shuffle per sub-step (chain1q_run)
        if (tid < 3) {
            Chain1Regs r = chain1_load(a, a.st, tid); r.ke = s_ke[tid];
            if (L.c1_quirk) chain1q_run(a, r, a.st, true, nullptr, tid);
            else chain1_run(a, r, a.st, true, nullptr, tid);
        }
        if (tid == 64) st[L.off_kesum] = 0.5 * (s_ke[0] + s_ke[2]);                  // Ref :586-588 (cached KE)
    } else if (tid == 0) {
        switch (L.C) {
            case 1: run_dualnh<1>(a, a.st, a.st, true, nullptr, s_chain, s_ke[0], s_ke[1], s_ke[2]); break;
            case 2: run_dualnh<2>(a, a.st, a.st, true, nullptr, s_chain, s_ke[0], s_ke[1], s_ke[2]); break;
            case 3: run_dualnh<3>(a, a.st, a.st, true, nullptr, s_chain, s_ke[0], s_ke[1], s_ke[2]); break;
            case 4: run_dualnh<4>(a, a.st, a.st, true, nullptr, s_chain, s_ke[0], s_ke[1], s_ke[2]); break;
            default: run_dualnh<0>(a, a.st, a.st, true, nullptr, s_chain, s_ke[0], s_ke[1], s_ke[2]); break;               // host checked 4*(2C+2) <= CHAIN_LDS_DOUBLES
        }
    }
    CHAIN_TRACE(2);
}
// Chains of 5-16 links (TGNH; ten is the reference test's value, TestReferenceDrudeTGNHIntegrator.cpp:166): a kernel per chain
// length with the links of a thermostat in the REGISTERS of its lane, every loop unrolled -- run_tgnh<CC>, the form of the chains
// of 2-4 links (no range test inside the loop, the exponentials that repeat within a sub-step taken once, the wide form when an
// equilibrating box leaves the short polynomial's range).  A thermostat's half step is 2 C S link updates, each waiting for the one
// before (the sweeps of Cu :566-571 / :586-592 bounce from end to end), run by one wavefront that issues one fp64 instruction per
// ~8 cycles: what it costs is instructions per update.  A link per lane (chain_lanes_run, round 3) pays per update two DPP moves
// for the neighbour's value, a range-tested exponential in EVERY lane and four conditional moves to commit in one: ~340
// instructions per sub-step of ten links against ~150 here.  Kernels of their own so that chain_kernel's code stays what it was
// (a launch starts with a cold instruction cache); a thermostat per lane, the Drude thermostat on the second wavefront.
template <int CC>
__global__ __launch_bounds__(BLOCK) void chain_long_kernel(const ChainArgs a) {
    __shared__ double sred[BLOCK / 64][MAX_GROUPS + 2];
    __shared__ double s_chain[2 * XCHG_MAX_WORLD * XCHG_NT_PAD > 64 ? 2 * XCHG_MAX_WORLD * XCHG_NT_PAD : 64];
    __shared__ double s_ke[MAX_GROUPS + 2];
    const ChainLayout& L = a.L;
    const int NT = L.NT, tid = threadIdx.x;
    chain_prologue(a, sred, s_chain, s_ke);
    if (!a.do_sum) __syncthreads();
    int itg = -1;
    if (tid < NT - 1) itg = tid;
    else if (tid == 64) itg = NT - 1;
    if (itg >= 0) run_tgnh<CC>(a, a.st, a.st, true, nullptr, itg, nullptr, s_ke[itg]);
    if (tid == 0) {                                                  // Cu :493-497
        double s = 0.0;
        for (int i = 0; i < NT; i++) s += s_ke[i];
        a.st[L.off_kesum] = 0.5 * s;
    }
}
// dualNH, chains of 5-16 links: ten links WITHOUT useDrudeNHChains are the values of the reference's own test
// (TestReferenceDrudeTGNHIntegrator.cpp:166), i.e. the coupled chain of Ref :476-503 with eleven moving entries.  chain_kernel runs
// such chains as the transcription on LDS-resident vectors (run_dualnh<0>: every access a ~100-cycle round trip on a path that is
// serial by nature); here the entries live in one lane's REGISTERS and the exponentials are the fast chains' polynomials
// (dualnh_quirk_fast<CC>), and with useDrudeNHChains the two independent chains take a lane each (run_dualnh_pair<CC>: lanes 0
// and 2 through chain_both_fast).  The transcription stays behind as what runs when an argument leaves the polynomials' range.
template <int CC>
__global__ __launch_bounds__(BLOCK) void chain_dualnh_long_kernel(const ChainArgs a) {
    __shared__ double sred[BLOCK / 64][MAX_GROUPS + 2];
    __shared__ double s_chain[CHAIN_LDS_DOUBLES];
    __shared__ double s_ke[MAX_GROUPS + 2];
    const ChainLayout& L = a.L;
    const int tid = threadIdx.x;
    chain_prologue(a, sred, s_chain, s_ke);
    if (!a.do_sum) __syncthreads();
    if (tid >= 64) return;
    bool done = false;
    if (L.use_drude_chains != 0) {
        bool ok = false;
        if (tid == 0 || tid == 2) ok = run_dualnh_pair<CC>(a, a.st, a.st, true, nullptr, tid, s_ke[0], s_ke[1], s_ke[2]);
        done = __shfl((int)ok, 0, 64) != 0;                          // (the same answer in both lanes: chain_fast votes)
    } else if (tid == 0) {
        done = dualnh_quirk_fast<CC, true>(a, a.st, a.st, true, nullptr, s_ke[0], s_ke[1], s_ke[2]);
    }
    if (!done && tid == 0) run_dualnh<0, true, false>(a, a.st, a.st, true, nullptr, s_chain, s_ke[0], s_ke[1], s_ke[2]);
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------
// big_com_kernel: COM velocity of the molecules longer than a tile (K :82-113 for those), one work-group each.
// kick = 1 gives the COM after the half kick that the KE launch is about to apply: sum m v' = sum (m v + dt/2 F).
// Also leaves the molecule's M v_com^2 (K :152-158) in its own partial row.
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void big_com_kernel(const BigComArgs a) {
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::mixed4 mixed4;
    __shared__ double sred[BLOCK / 64][4];
    const mixed4* __restrict__ velm = reinterpret_cast<const mixed4*>(a.velm);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const mixed fscale = (mixed)(0.5 * a.dt / 4294967296.0);
    for (int b = blockIdx.x; b < a.n; b += gridDim.x) {
        const int2 rt = a.table[b];
        double sx = 0, sy = 0, sz = 0, sm = 0;
        for (int j = tid; j < rt.x; j += BLOCK) {
            const int i = rt.y + j;
            const mixed4 v = velm[i];
            if (v.w != 0) {
                mixed vx = v.x, vy = v.y, vz = v.z;
                if (a.kick) {
                    const mixed c = fscale * v.w;
                    vx += c * (mixed)a.force[i]; vy += c * (mixed)a.force[i + a.padded]; vz += c * (mixed)a.force[i + 2 * a.padded];
                }
                const mixed m = rcp_(v.w);
                sx += (double)(vx * m); sy += (double)(vy * m); sz += (double)(vz * m); sm += (double)m;
            }
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); sm = wave_sum(sm);
        if (lane == 0) { sred[wv][0] = sx; sred[wv][1] = sy; sred[wv][2] = sz; sred[wv][3] = sm; }
        __syncthreads();
        if (tid == 0) {
            double x = 0, y = 0, z = 0, m = 0;
            for (int w = 0; w < BLOCK / 64; w++) { x += sred[w][0]; y += sred[w][1]; z += sred[w][2]; m += sred[w][3]; }
            const double wi = 1.0 / m;
            x *= wi; y *= wi; z *= wi;
            reinterpret_cast<mixed4*>(a.big_com)[b] = mk4((mixed)x, (mixed)y, (mixed)z, (mixed)wi);
            for (int k = 0; k < a.NT; k++) a.partials[(size_t)b * a.NT + k] = 0.0;
            const mixed cx = (mixed)x, cy = (mixed)y, cz = (mixed)z, cw = (mixed)wi;     // as the tiles will read it
            a.partials[(size_t)b * a.NT + a.G] = ((double)cx * cx + (double)cy * cy + (double)cz * cz) / (double)cw;
        }
        __syncthreads();
    }
}

hipError_t launch_big_com(int precision, const BigComArgs& a, hipStream_t s) {
    int grid = a.n < 1 ? 1 : (a.n > 1024 ? 1024 : a.n);
    switch (precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((big_com_kernel<TGNH_PREC_SINGLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((big_com_kernel<TGNH_PREC_MIXED>), dim3(grid), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_DOUBLE: TGNH_LAUNCH((big_com_kernel<TGNH_PREC_DOUBLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// harness force (bench/test workload; not part of the reference)
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void force_kernel(const ForceArgs a) {
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed mixed;
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    const real4* __restrict__ x0 = reinterpret_cast<const real4*>(a.x0);      // (site x,y,z ; w = 1 if tethered)
    const mixed kd = (mixed)a.k_drude, kt = (mixed)a.k_tether;
    const int lane = threadIdx.x & 63;
    // uniform trip count per wavefront so the shuffles below see all 64 lanes
    const int nround = (a.n + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    for (int rr = 0; rr < nround; rr++) {
        // optionally last chunk first (lane order inside a chunk unchanged): start where the previous launch ended
        const int r = a.reverse ? nround - 1 - rr : rr;
        const int blk = a.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
        const int i = (r * gridDim.x + blk) * BLOCK + threadIdx.x;
        const bool in = i < a.n;
        uint32_t m = 0;
        mixed x = 0, y = 0, z = 0;
        if (in) {
            m = a.meta[i];
            const real4 p = posq[i];
            x = p.x; y = p.y; z = p.z;
            if (PREC == TGNH_PREC_MIXED) { const float4 c = pcorr[i]; x += (mixed)c.x; y += (mixed)c.y; z += (mixed)c.z; }
        }
        const uint32_t role = m & 3u;
        const int off = (int)((m >> 10) & 2047u) - 1024;
        // partner position: from the partner's lane when it is in this wavefront (the usual case: partners are
        // neighbours), else one more global read
        const int pl = lane + off;
        const int src = (pl >= 0 && pl < 64) ? pl : lane;
        mixed ox = __shfl(x, src, 64), oy = __shfl(y, src, 64), oz = __shfl(z, src, 64);
        mixed fx = 0, fy = 0, fz = 0;
        if (in) {
            if (role != ROLE_DRUDE) {                                   // tether of massive non-Drude sites
                const real4 s = x0[i];
                if (s.w != 0) { fx = -kt * (x - (mixed)s.x); fy = -kt * (y - (mixed)s.y); fz = -kt * (z - (mixed)s.z); }
            }
            if (role != ROLE_NORMAL) {                                  // Drude spring
                if (src != pl) {
                    const int j = i + off;
                    const real4 q = posq[j];
                    ox = q.x; oy = q.y; oz = q.z;
                    if (PREC == TGNH_PREC_MIXED) { const float4 c = pcorr[j]; ox += (mixed)c.x; oy += (mixed)c.y; oz += (mixed)c.z; }
                }
                // separation Drude - parent; force -k sep on the Drude, +k sep on the parent
                const bool is_d = role == ROLE_DRUDE;
                const mixed sgn = is_d ? (mixed)-1 : (mixed)1;
                const mixed sx = is_d ? x - ox : ox - x, sy = is_d ? y - oy : oy - y, sz = is_d ? z - oz : oz - z;
                fx += sgn * kd * sx; fy += sgn * kd * sy; fz += sgn * kd * sz;
            }
            a.force[i] = (long long)(fx * (mixed)4294967296.0);
            a.force[i + a.padded] = (long long)(fy * (mixed)4294967296.0);
            a.force[i + 2 * a.padded] = (long long)(fz * (mixed)4294967296.0);
        }
    }
}

// ---------------------------------------------------------------------------
// plain / time-shifted kinetic energy (A12): 1/2 sum (v + F ts /m)^2 m
// Cu :656 (ts = 0) ; Ref :70-98 (ts = dt/2, no constraints)
// ---------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(BLOCK) void plain_ke_kernel(const void* velm_, const long long* force, int n, int padded,
                                                         double ts, double* out) {
    typedef typename Prec<PREC>::mixed4 mixed4;
    __shared__ double sred[BLOCK / 64];
    const mixed4* __restrict__ velm = reinterpret_cast<const mixed4*>(velm_);
    const double fs = ts / 4294967296.0;
    double e = 0.0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const mixed4 v = velm[i];
        if (v.w != 0) {
            double vx = v.x, vy = v.y, vz = v.z;
            if (ts != 0.0) {
                const double c = fs * (double)v.w;
                vx += c * (double)force[i]; vy += c * (double)force[i + padded]; vz += c * (double)force[i + 2 * padded];
            }
            e += (vx * vx + vy * vy + vz * vz) / (double)v.w;
        }
    }
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < BLOCK / 64; w++) s += sred[w];
        out[1 + blockIdx.x] = s;                 // one partial per work-group: no atomics, the order of the sum is fixed below
    }
}

// out[0] = 1/2 sum of the nparts work-group partials out[1 ..], in index order: the query is reproducible bit for bit
__global__ __launch_bounds__(BLOCK) void plain_ke_sum_kernel(double* out, int nparts) {
    __shared__ double sred[BLOCK / 64];
    double e = 0.0;
    for (int i = threadIdx.x; i < nparts; i += BLOCK) e += out[1 + i];
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < BLOCK / 64; w++) s += sred[w];
        out[0] = 0.5 * s;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
typedef void (*tile_fn_t)(const TileArgs);

template <int PREC, int OPS>
static tile_fn_t tile_fn_gb(int gb) {
    if constexpr ((OPS & OP_KE) != 0) {
        if (gb == 0) return tile_kernel<PREC, OPS, 0>;
        if (gb <= 1) return tile_kernel<PREC, OPS, 1>;
        if (gb <= 4) return tile_kernel<PREC, OPS, 4>;
        return tile_kernel<PREC, OPS, 8>;
    } else {
        return tile_kernel<PREC, OPS, 1>;
    }
}

template <int PREC>
static tile_fn_t tile_fn_ops(int ops, int gb) {
    switch (ops) {
        case OP_KE: return tile_fn_gb<PREC, OP_KE>(gb);
        case OP_SCALE: return tile_fn_gb<PREC, OP_SCALE>(gb);
        case OP_SCALE | OP_KICK | OP_DRIFT: return tile_fn_gb<PREC, OP_SCALE | OP_KICK | OP_DRIFT>(gb);
        case OP_KICK | OP_KE: return tile_fn_gb<PREC, OP_KICK | OP_KE>(gb);
        case OP_KICK | OP_KE | OP_NOSTORE: return tile_fn_gb<PREC, OP_KICK | OP_KE | OP_NOSTORE>(gb);
        case OP_PREKICK | OP_SCALE | OP_KICK | OP_DRIFT: return tile_fn_gb<PREC, OP_PREKICK | OP_SCALE | OP_KICK | OP_DRIFT>(gb);
        case OP_PREKICK | OP_SCALE: return tile_fn_gb<PREC, OP_PREKICK | OP_SCALE>(gb);
        case OP_KICK: return tile_fn_gb<PREC, OP_KICK>(gb);
        case OP_SCALE | OP_KICK | OP_POSDELTA: return tile_fn_gb<PREC, OP_SCALE | OP_KICK | OP_POSDELTA>(gb);
        case OP_MOVE: return tile_fn_gb<PREC, OP_MOVE>(gb);
        default: return nullptr;
    }
}

static tile_fn_t tile_fn(int precision, int ops, int gb) {
    switch (precision) {
        case TGNH_PREC_SINGLE: return tile_fn_ops<TGNH_PREC_SINGLE>(ops, gb);
        case TGNH_PREC_MIXED: return tile_fn_ops<TGNH_PREC_MIXED>(ops, gb);
        case TGNH_PREC_DOUBLE: return tile_fn_ops<TGNH_PREC_DOUBLE>(ops, gb);
        default: return nullptr;
    }
}

// the instantiations whose in-kernel chain may have 2-4 links (rescale launches only; no KE bins in any of them)
template <int PREC> static tile_fn_t tile_fn_multi(int ops) {
    switch (ops) {
        case OP_SCALE: return tile_kernel<PREC, OP_SCALE, 1, true>;
        case OP_SCALE | OP_KICK | OP_DRIFT: return tile_kernel<PREC, OP_SCALE | OP_KICK | OP_DRIFT, 1, true>;
        case OP_PREKICK | OP_SCALE | OP_KICK | OP_DRIFT: return tile_kernel<PREC, OP_PREKICK | OP_SCALE | OP_KICK | OP_DRIFT, 1, true>;
        case OP_PREKICK | OP_SCALE: return tile_kernel<PREC, OP_PREKICK | OP_SCALE, 1, true>;
        case OP_SCALE | OP_KICK | OP_POSDELTA: return tile_kernel<PREC, OP_SCALE | OP_KICK | OP_POSDELTA, 1, true>;
        default: return nullptr;
    }
}
static tile_fn_t tile_fn_any(int precision, int ops, int gb, bool multi) {
    if (!multi) return tile_fn(precision, ops, gb);
    switch (precision) {
        case TGNH_PREC_SINGLE: return tile_fn_multi<TGNH_PREC_SINGLE>(ops);
        case TGNH_PREC_MIXED: return tile_fn_multi<TGNH_PREC_MIXED>(ops);
        case TGNH_PREC_DOUBLE: return tile_fn_multi<TGNH_PREC_DOUBLE>(ops);
        default: return nullptr;
    }
}

hipError_t launch_tile(int precision, int ops, int gb, const TileArgs& a, int grid, size_t lds, hipStream_t s) {
    tile_fn_t fn = tile_fn_any(precision, ops, gb, a.chain_on && a.chain.L.C > 1);
    if (!fn) return hipErrorInvalidValue;
    TGNH_LAUNCH(fn, dim3(grid), dim3(TBLOCK), lds, s, a);
    return hipGetLastError();
}

template <int PREC, int OPS> static tile_fn_t wke_fn_gb(int gb) {
    if (gb <= 1) return wke_kernel<PREC, OPS, 1>;
    if (gb <= 4) return wke_kernel<PREC, OPS, 4>;
    return wke_kernel<PREC, OPS, 8>;
}
template <int PREC> static tile_fn_t wke_fn_ops(int ops, int gb) {
    switch (ops) {
        case OP_KE: return wke_fn_gb<PREC, OP_KE>(gb);
        case OP_KICK | OP_KE: return wke_fn_gb<PREC, OP_KICK | OP_KE>(gb);
        case OP_KICK | OP_KE | OP_NOSTORE: return wke_fn_gb<PREC, OP_KICK | OP_KE | OP_NOSTORE>(gb);
        default: return nullptr;
    }
}
static tile_fn_t wke_fn(int precision, int ops, int gb) {
    if (gb == 0) return nullptr;                          // more than 8 groups: LDS bins, the tile kernel
    switch (precision) {
        case TGNH_PREC_SINGLE: return wke_fn_ops<TGNH_PREC_SINGLE>(ops, gb);
        case TGNH_PREC_MIXED: return wke_fn_ops<TGNH_PREC_MIXED>(ops, gb);
        case TGNH_PREC_DOUBLE: return wke_fn_ops<TGNH_PREC_DOUBLE>(ops, gb);
        default: return nullptr;
    }
}
hipError_t launch_wke(int precision, int ops, int gb, const TileArgs& a, int grid, hipStream_t s) {
    tile_fn_t fn = wke_fn(precision, ops, gb);
    if (!fn) return hipErrorInvalidValue;
    TGNH_LAUNCH(fn, dim3(grid), dim3(TBLOCK), 0, s, a);
    return hipGetLastError();
}
int wke_blocks_per_cu(int precision, int ops, int gb) {
    tile_fn_t fn = wke_fn(precision, ops, gb);
    int n = 0;
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(fn), TBLOCK, 0) != hipSuccess) return 0;
    return n;
}

typedef void (*step_fn_t)(const TileArgs);
template <int PREC, int KIND> static step_fn_t step_fn_gb(int gb) {
    if (gb <= 1) return step_kernel<PREC, 1, KIND>;
    if (gb <= 4) return step_kernel<PREC, 4, KIND>;
    return step_kernel<PREC, 8, KIND>;
}
template <int PREC> static step_fn_t step_fn_kind(int kind, int gb) {
    switch (kind) {
        case STEP_DEFER: return step_fn_gb<PREC, STEP_DEFER>(gb);
        case STEP_PLAIN_BEGIN: return step_fn_gb<PREC, STEP_PLAIN_BEGIN>(gb);
        case STEP_PLAIN_END: return step_fn_gb<PREC, STEP_PLAIN_END>(gb);
        case STEP_SPLIT_BEGIN: return step_fn_gb<PREC, STEP_SPLIT_BEGIN>(gb);
        case STEP_SPLIT_END: return step_fn_gb<PREC, STEP_SPLIT_END>(gb);
        default: return nullptr;
    }
}
static step_fn_t step_fn(int precision, int gb, int kind) {
    if (gb == 0) return nullptr;                          // more than 8 groups: the tile kernels
    switch (precision) {
        case TGNH_PREC_SINGLE: return step_fn_kind<TGNH_PREC_SINGLE>(kind, gb);
        case TGNH_PREC_MIXED: return step_fn_kind<TGNH_PREC_MIXED>(kind, gb);
        case TGNH_PREC_DOUBLE: return step_fn_kind<TGNH_PREC_DOUBLE>(kind, gb);
        default: return nullptr;
    }
}
template <int PREC, bool MULTI> static step_fn_t wstep_fn_gb(int gb) {
    return gb <= 1 ? wstep_kernel<PREC, 1, MULTI> : gb <= 4 ? wstep_kernel<PREC, 4, MULTI> : wstep_kernel<PREC, 8, MULTI>;
}
static step_fn_t wstep_fn(int precision, int gb, bool multi) {
    if (gb == 0) return nullptr;
    switch (precision) {
        case TGNH_PREC_SINGLE: return multi ? wstep_fn_gb<TGNH_PREC_SINGLE, true>(gb) : wstep_fn_gb<TGNH_PREC_SINGLE, false>(gb);
        case TGNH_PREC_MIXED: return multi ? wstep_fn_gb<TGNH_PREC_MIXED, true>(gb) : wstep_fn_gb<TGNH_PREC_MIXED, false>(gb);
        case TGNH_PREC_DOUBLE: return multi ? wstep_fn_gb<TGNH_PREC_DOUBLE, true>(gb) : wstep_fn_gb<TGNH_PREC_DOUBLE, false>(gb);
        default: return nullptr;
    }
}
hipError_t launch_wstep(int precision, int gb, bool multi, const TileArgs& a, int grid, hipStream_t s) {
    step_fn_t fn = wstep_fn(precision, gb, multi);
    if (!fn) return hipErrorInvalidValue;
    TGNH_LAUNCH(fn, dim3(grid), dim3(WBLOCK), 0, s, a);
    return hipGetLastError();
}
int wstep_blocks_per_cu(int precision, int gb, bool multi) {
    step_fn_t fn = wstep_fn(precision, gb, multi);
    int n = 0;
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(fn), WBLOCK, 0) != hipSuccess) return 0;
    return n;
}
int step_kind_ops2(int kind) { return step_ops2(kind); }
hipError_t launch_step(int precision, int gb, int kind, const TileArgs& a, int grid, size_t lds, hipStream_t s) {
    step_fn_t fn = step_fn(precision, gb, kind);
    if (!fn) return hipErrorInvalidValue;
    TGNH_LAUNCH(fn, dim3(grid), dim3(TBLOCK), lds, s, a);
    return hipGetLastError();
}
int step_blocks_per_cu(int precision, int gb, int kind, size_t lds) {
    step_fn_t fn = step_fn(precision, gb, kind);
    int n = 0;
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(fn), TBLOCK, lds) != hipSuccess) return 0;
    return n;
}

int tile_blocks_per_cu(int precision, int ops, int gb, size_t lds, bool multi) {
    tile_fn_t fn = tile_fn_any(precision, ops, gb, multi);
    int n = 0;
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(fn), TBLOCK, lds) != hipSuccess) return 0;
    return n;
}

hipError_t launch_chain(const ChainArgs& a, hipStream_t s) {
    if (!a.do_chain) TGNH_LAUNCH(rowsum_kernel, dim3(1), dim3(BLOCK), 0, s, a);
    else if (a.L.mode == TGNH_MODE_TGNH && a.L.C > 4 && a.L.C <= 16 && !a.lanes) {
        switch (a.L.C) {
#define TGNH_LONG(c) case c: TGNH_LAUNCH(chain_long_kernel<c>, dim3(1), dim3(BLOCK), 0, s, a); break;
            TGNH_LONG(5) TGNH_LONG(6) TGNH_LONG(7) TGNH_LONG(8) TGNH_LONG(9) TGNH_LONG(10) TGNH_LONG(11) TGNH_LONG(12)
            TGNH_LONG(13) TGNH_LONG(14) TGNH_LONG(15) TGNH_LONG(16)
#undef TGNH_LONG
        }
    }
    else if (a.L.mode == TGNH_MODE_DUALNH && a.L.C > 4 && a.L.C <= 16) {
        switch (a.L.C) {
#define TGNH_DLONG(c) case c: TGNH_LAUNCH(chain_dualnh_long_kernel<c>, dim3(1), dim3(BLOCK), 0, s, a); break;
            TGNH_DLONG(5) TGNH_DLONG(6) TGNH_DLONG(7) TGNH_DLONG(8) TGNH_DLONG(9) TGNH_DLONG(10) TGNH_DLONG(11) TGNH_DLONG(12)
            TGNH_DLONG(13) TGNH_DLONG(14) TGNH_DLONG(15) TGNH_DLONG(16)
#undef TGNH_DLONG
        }
    }
    else TGNH_LAUNCH(chain_kernel, dim3(1), dim3(BLOCK), 0, s, a);
#ifdef TGNH_TUNING
    // timing experiment only (the thermostat advances twice): the same launch again, its code now in the caches
    static const int again = getenv("TGNH_CHAIN_REPEAT") ? atoi(getenv("TGNH_CHAIN_REPEAT")) : 0;
    for (int r = 0; r < again && a.do_chain; r++) { ChainArgs b = a; b.do_sum = 0; b.commit = 0; b.x_send = 0; b.x_wait = 0; TGNH_LAUNCH(chain_kernel, dim3(1), dim3(BLOCK), 0, s, b); }
#endif
    return hipGetLastError();
}

// The same forces from the packed sites (ForceArgs::sflag): every load of a slot is issued before the first is used, the
// site comes from the compact array (its index: the chunk's base + the tethered lanes before this one), the meta word is
// read only by a slot whose partner is more than 15 slots away.
template <int PREC>
__global__ __launch_bounds__(BLOCK) void force_packed_kernel(const ForceArgs a) {
    typedef typename Prec<PREC>::real real;
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed mixed;
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    const real* __restrict__ sites = reinterpret_cast<const real*>(a.sites);
    const mixed kd = (mixed)a.k_drude, kt = (mixed)a.k_tether;
    const int lane = threadIdx.x & 63;
    const int nround = (a.n + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    for (int rr = 0; rr < nround; rr++) {
        const int r = a.reverse ? nround - 1 - rr : rr;
        const int blk = a.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
        const int i = (r * gridDim.x + blk) * BLOCK + threadIdx.x;           // (a wavefront covers one aligned 64-slot chunk)
        const bool in = i < a.n;
        uint32_t b = 0;
        real4 p = {}; float4 c = {};
        if (in) {
            b = a.sflag[i];
            p = posq[i];
            if (PREC == TGNH_PREC_MIXED) c = pcorr[i];
        }
        const bool tethered = (b & 4u) != 0;
        const unsigned long long before = __ballot(tethered) & ((1ull << lane) - 1ull);
        real s0 = 0, s1 = 0, s2 = 0;
        if (tethered) {
            const real* rec = sites + 3 * (size_t)(a.sbase[i >> 6] + (uint32_t)__popcll(before));
            s0 = rec[0]; s1 = rec[1]; s2 = rec[2];
        }
        mixed x = p.x, y = p.y, z = p.z;
        if (PREC == TGNH_PREC_MIXED) { x += (mixed)c.x; y += (mixed)c.y; z += (mixed)c.z; }
        const uint32_t role = b & 3u;
        int off = (int)(b >> 3) - 16;
        if (in && (b >> 3) == 0) off = (int)((a.meta[i] >> 10) & 2047u) - 1024;
        const int pl = lane + off;
        const int src = (pl >= 0 && pl < 64) ? pl : lane;
        mixed ox = __shfl(x, src, 64), oy = __shfl(y, src, 64), oz = __shfl(z, src, 64);
        mixed fx = 0, fy = 0, fz = 0;
        if (in) {
            if (tethered) { fx = -kt * (x - (mixed)s0); fy = -kt * (y - (mixed)s1); fz = -kt * (z - (mixed)s2); }
            if (role != ROLE_NORMAL) {
                if (src != pl) {
                    const int j = i + off;
                    const real4 q = posq[j];
                    ox = q.x; oy = q.y; oz = q.z;
                    if (PREC == TGNH_PREC_MIXED) { const float4 cq = pcorr[j]; ox += (mixed)cq.x; oy += (mixed)cq.y; oz += (mixed)cq.z; }
                }
                const bool is_d = role == ROLE_DRUDE;
                const mixed sgn = is_d ? (mixed)-1 : (mixed)1;
                const mixed sx = is_d ? x - ox : ox - x, sy = is_d ? y - oy : oy - y, sz = is_d ? z - oz : oz - z;
                fx += sgn * kd * sx; fy += sgn * kd * sy; fz += sgn * kd * sz;
            }
            a.force[i] = (long long)(fx * (mixed)4294967296.0);
            a.force[i + a.padded] = (long long)(fy * (mixed)4294967296.0);
            a.force[i + 2 * a.padded] = (long long)(fz * (mixed)4294967296.0);
        }
    }
}

// ... and from LATTICE sites (ForceArgs::lat_*: one molecule repeated on a simple cubic lattice, checked slot by slot by
// tgnh_harness_pack_sites): flag byte and site are functions of the slot index -- molecule m = i div k, slot i - m k of it, lattice
// point (m div side^2, (m div side) mod side, m mod side) -- so the kernel reads positions and writes forces, nothing else: the
// 56 B per slot (mixed) the call-out cannot do without.  The site is fl(fl64(index x spacing) + geom), two roundings and a
// conversion, the bits numpy gave the packed sites (__dmul_rn / __dadd_rn: never contracted; the force arithmetic itself is the
// packed kernel's, under the same contraction rules: the same forces bit for bit); x div d as floor((x + 1/2) / d): never within
// rounding of an integer.
template <int PREC>
__global__ __launch_bounds__(BLOCK) void force_lattice_kernel(const ForceArgs a) {
    typedef typename Prec<PREC>::real real;
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed mixed;
    __shared__ double s_geom[64 * 3];
    __shared__ unsigned char s_flag[64];
    const real4* __restrict__ posq = reinterpret_cast<const real4*>(a.posq);
    const float4* __restrict__ pcorr = reinterpret_cast<const float4*>(a.posq_corr);
    if (threadIdx.x < 64) s_flag[threadIdx.x] = a.lat_tab[threadIdx.x];
    if (threadIdx.x < 192) s_geom[threadIdx.x] = reinterpret_cast<const double*>(a.lat_tab + 64)[threadIdx.x];
    __syncthreads();
    const mixed kd = (mixed)a.k_drude, kt = (mixed)a.k_tether;
    const int lane = threadIdx.x & 63;
    const int nround = (a.n + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    for (int rr = 0; rr < nround; rr++) {
        const int r = a.reverse ? nround - 1 - rr : rr;
        const int blk = a.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
        const int i = (r * gridDim.x + blk) * BLOCK + threadIdx.x;
        const bool in = i < a.n;
        real4 p = {}; float4 c = {};
        if (in) {
            p = posq[i];
            if (PREC == TGNH_PREC_MIXED) c = pcorr[i];
        }
        const int mloc = (int)(((double)i + 0.5) * a.lat_inv_k), pos = in ? i - mloc * a.lat_k : 0;
        const int mol = a.lat_mol0 + mloc;                               // this handle's molecules start at lat_mol0 of the box (shards)
        const uint32_t b = in ? s_flag[pos] : 0u;
        const bool tethered = (b & 4u) != 0;
        real s0 = 0, s1 = 0, s2 = 0;
        if (tethered) {
            const int ix = (int)(((double)mol + 0.5) * a.lat_inv_side2), rem = mol - ix * a.lat_side * a.lat_side;
            const int iy = (int)(((double)rem + 0.5) * a.lat_inv_side), iz = rem - iy * a.lat_side;
            s0 = (real)__dadd_rn(__dmul_rn((double)ix, a.lat_spacing), s_geom[3 * pos]);
            s1 = (real)__dadd_rn(__dmul_rn((double)iy, a.lat_spacing), s_geom[3 * pos + 1]);
            s2 = (real)__dadd_rn(__dmul_rn((double)iz, a.lat_spacing), s_geom[3 * pos + 2]);
        }
        mixed x = p.x, y = p.y, z = p.z;
        if (PREC == TGNH_PREC_MIXED) { x += (mixed)c.x; y += (mixed)c.y; z += (mixed)c.z; }
        const uint32_t role = b & 3u;
        const int off = (int)(b >> 3) - 16;
        const int pl = lane + off;
        const int src = (pl >= 0 && pl < 64) ? pl : lane;
        mixed ox = __shfl(x, src, 64), oy = __shfl(y, src, 64), oz = __shfl(z, src, 64);
        mixed fx = 0, fy = 0, fz = 0;
        if (in) {
            if (tethered) { fx = -kt * (x - (mixed)s0); fy = -kt * (y - (mixed)s1); fz = -kt * (z - (mixed)s2); }
            if (role != ROLE_NORMAL) {
                if (src != pl) {
                    const int j = i + off;
                    const real4 q = posq[j];
                    ox = q.x; oy = q.y; oz = q.z;
                    if (PREC == TGNH_PREC_MIXED) { const float4 cq = pcorr[j]; ox += (mixed)cq.x; oy += (mixed)cq.y; oz += (mixed)cq.z; }
                }
                const bool is_d = role == ROLE_DRUDE;
                const mixed sgn = is_d ? (mixed)-1 : (mixed)1;
                const mixed sx = is_d ? x - ox : ox - x, sy = is_d ? y - oy : oy - y, sz = is_d ? z - oz : oz - z;
                fx += sgn * kd * sx; fy += sgn * kd * sy; fz += sgn * kd * sz;
            }
            a.force[i] = (long long)(fx * (mixed)4294967296.0);
            a.force[i + a.padded] = (long long)(fy * (mixed)4294967296.0);
            a.force[i + 2 * a.padded] = (long long)(fz * (mixed)4294967296.0);
        }
    }
}

hipError_t launch_force(int precision, const ForceArgs& a, hipStream_t s) {
    int grid = (a.n + BLOCK - 1) / BLOCK;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (a.lat_tab) {
        switch (precision) {
            case TGNH_PREC_SINGLE: TGNH_LAUNCH((force_lattice_kernel<TGNH_PREC_SINGLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
            case TGNH_PREC_MIXED: TGNH_LAUNCH((force_lattice_kernel<TGNH_PREC_MIXED>), dim3(grid), dim3(BLOCK), 0, s, a); break;
            case TGNH_PREC_DOUBLE: TGNH_LAUNCH((force_lattice_kernel<TGNH_PREC_DOUBLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.sflag) {
        switch (precision) {
            case TGNH_PREC_SINGLE: TGNH_LAUNCH((force_packed_kernel<TGNH_PREC_SINGLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
            case TGNH_PREC_MIXED: TGNH_LAUNCH((force_packed_kernel<TGNH_PREC_MIXED>), dim3(grid), dim3(BLOCK), 0, s, a); break;
            case TGNH_PREC_DOUBLE: TGNH_LAUNCH((force_packed_kernel<TGNH_PREC_DOUBLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((force_kernel<TGNH_PREC_SINGLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((force_kernel<TGNH_PREC_MIXED>), dim3(grid), dim3(BLOCK), 0, s, a); break;
        case TGNH_PREC_DOUBLE: TGNH_LAUNCH((force_kernel<TGNH_PREC_DOUBLE>), dim3(grid), dim3(BLOCK), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_plain_ke(int precision, const void* velm, const long long* force, int n, int padded,
                           double time_shift, double* out, hipStream_t s) {
    int grid = (n + BLOCK - 1) / BLOCK;
    if (grid > PLAIN_KE_PARTS) grid = PLAIN_KE_PARTS;
    if (grid < 1) grid = 1;
    switch (precision) {
        case TGNH_PREC_SINGLE: TGNH_LAUNCH((plain_ke_kernel<TGNH_PREC_SINGLE>), dim3(grid), dim3(BLOCK), 0, s, velm, force, n, padded, time_shift, out); break;
        case TGNH_PREC_MIXED: TGNH_LAUNCH((plain_ke_kernel<TGNH_PREC_MIXED>), dim3(grid), dim3(BLOCK), 0, s, velm, force, n, padded, time_shift, out); break;
        case TGNH_PREC_DOUBLE: TGNH_LAUNCH((plain_ke_kernel<TGNH_PREC_DOUBLE>), dim3(grid), dim3(BLOCK), 0, s, velm, force, n, padded, time_shift, out); break;
        default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    TGNH_LAUNCH(plain_ke_sum_kernel, dim3(1), dim3(BLOCK), 0, s, out, grid);
    return hipGetLastError();
}

}  // namespace tgnh

#ifdef TGNH_TRACE
extern "C" int tgnh_debug_read_chain_dbg(double* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnh::g_chain_dbg), sizeof(double) * 4);
}
#endif
