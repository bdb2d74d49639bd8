// tgnh_host.cpp -- host side of the C ABI (include/drude_tgnh.h): topology and
// tile construction (A1), degrees of freedom and thermostat masses (A2), step
// orchestration (A11) and the queries.  Kernels live in tgnh_kernels.hip.
//
// Reference semantics followed (scychon/openmm_drudeNose):
//   Ref = platforms/reference/src/ReferenceDrudeTGNHKernels.cpp
//   Cu  = platforms/cuda/src/CudaDrudeTGNHKernels.cpp
//   API = openmmapi/src/DrudeTGNHIntegrator.cpp
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>

#include <dlfcn.h>
#include <mutex>
// RCCL is bound with dlopen at run time (rccl() below); of its header only five types and three enumerators are used.  A ROCm
// install without the RCCL headers still builds this library (and the OpenMM glue): the declarations below are RCCL's ABI
// (rccl.h: ncclUniqueId is 128 bytes, ncclSuccess = 0, ncclSum = 0, ncclFloat64 = ncclDouble = 8).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
}
#endif

#include "tgnh_internal.h"
#ifndef TGNH_MEETING_MEM
#define TGNH_MEETING_MEM hipDeviceMallocFinegrained
#endif

using namespace tgnh;

static thread_local std::string g_err;
extern "C" const char* tgnh_last_error(void) { return g_err.c_str(); }
extern "C" int tgnh_abi_version(void) { return TGNH_ABI_VERSION; }

void tgnh_set_error(const std::string& msg) { g_err = msg; }

static tgnh_status fail(tgnh_status code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(TGNH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)
#define CHECK_H(h)                                                 \
    do {                                                           \
        if (!(h)) return fail(TGNH_ERR_ARG, "null handle");        \
    } while (0)

static void note_status(tgnh_handle h, uint32_t flags);
static tgnh_status entry(tgnh_handle h, bool need_bufs);
static tgnh_status flush_impl(tgnh_handle h, hipStream_t s);
static tgnh_status settle_kick(tgnh_handle h, hipStream_t s);
static tgnh_status settle_end(tgnh_handle h, hipStream_t s);
static bool resident_now(tgnh_handle h);

// ---------------------------------------------------------------------------
// A1 for the gather path (tgnh_gather.hip): the reference's index lists -- normalParticles, pairParticles (Ref :113-137,
// Cu :111-151), particleTempGroup, particleResId, particlesInResidues (Cu :114-125) -- turned per-particle (every particle's pair
// partner | is-Drude << 31, or -1; its group; its residue's index) so that the kernels walk the arrays once, in index order, plus
// the residue table (count, first) and nothing else: no tiles, no per-slot words.  Taken by build_topology for what the tiles
// cannot hold (c->generic_reason says what).
// ---------------------------------------------------------------------------
static tgnh_status build_gather_topology(tgnh_context* c, const std::vector<int>& role, const std::vector<int>& partner,
                                         const std::vector<int>& res_order) {
    const tgnh_desc& d = c->d;
    const int N = d.num_particles;
    const bool com = d.mode == TGNH_MODE_TGNH && d.use_com_temp_group;
    c->tile_start.assign(1, N); c->tile_res.assign(1, 0); c->num_tiles = 0;
    c->res_entries.assign(1, make_int2(0, 0));
    c->meta.clear(); c->wave_tile.clear(); c->wmeta.clear(); c->num_wtiles = 0;
    c->tile_pat.assign(1, 0u); c->wtile_pat.assign(1, 0u); c->pattern.assign(PATTERN_WORDS, 0u); c->wpattern.assign(PATTERN_WORDS, 0u);
    c->big_first.clear(); c->big_count.clear(); c->num_big = 0;
    // residues in order of first appearance; a particle's residue as its index in that table
    c->g_res_table.clear(); c->g_resid.assign(N, 0);
    if (com) {
        std::vector<int> internal(d.num_residues, -1);
        for (int r : res_order) { internal[r] = (int)c->g_res_table.size(); c->g_res_table.push_back(make_int2(c->res_count[r], c->res_first[r])); }
        for (int i = 0; i < N; i++) c->g_resid[i] = internal[c->resid[i]];
    }
    if (c->g_res_table.empty()) c->g_res_table.push_back(make_int2(0, 0));
    {   // gather_com_kernel's lanes per residue: the power of two that holds the mean residue
        const size_t mean = c->g_res_table.empty() ? 1 : ((size_t)N + c->g_res_table.size() - 1) / c->g_res_table.size();
        c->g_com_lanes = 1;
        while (c->g_com_lanes < 64 && (size_t)c->g_com_lanes < mean) c->g_com_lanes *= 2;
    }
    c->g_partner.assign(N, -1);                                  // the other member of a particle's pair | is-Drude << 31
    for (int i = 0; i < N; i++)
        if (partner[i] >= 0) c->g_partner[i] = role[i] == (int)ROLE_DRUDE ? (int)((unsigned)partner[i] | 0x80000000u) : partner[i];
    if (c->host_only) return TGNH_OK;
    auto up = [](auto** dst, const auto& v) -> hipError_t {
        typedef typename std::remove_reference<decltype(v)>::type::value_type T;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), sizeof(T) * std::max<size_t>(v.size(), 1));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
    };
    HIP_OK(up(&c->d_g_group, c->group));
    HIP_OK(up(&c->d_g_resid, c->g_resid));
    HIP_OK(up(&c->d_g_res_table, c->g_res_table));
    HIP_OK(up(&c->d_g_partner, c->g_partner));
    HIP_OK(hipMalloc(&c->d_g_com, 32 * c->g_res_table.size()));                 // mixed4 per residue
    HIP_OK(hipMemset(c->d_g_com, 0, 32 * c->g_res_table.size()));
    // (the tiled path's tables, so that nothing holds a null pointer; no launch of the gather path reads them)
    HIP_OK(hipMalloc(&c->d_meta, sizeof(uint32_t)));
    HIP_OK(hipMalloc(&c->d_tile_start, sizeof(int)));
    HIP_OK(hipMalloc(&c->d_tile_res, sizeof(int)));
    HIP_OK(hipMalloc(&c->d_res_table, sizeof(int2)));
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// A1: topology + tiles
// ---------------------------------------------------------------------------
static tgnh_status build_topology(tgnh_context* c, const tgnh_desc* d) {
    const int N = d->num_particles, P = d->num_pairs;
    const bool tg = d->mode == TGNH_MODE_TGNH;
    const bool com = tg && d->use_com_temp_group;
    c->mass.assign(d->mass, d->mass + N);
    c->pair_drude.assign(d->pair_drude, d->pair_drude + P);
    c->pair_parent.assign(d->pair_parent, d->pair_parent + P);
    // dualNH: the Reference platform never reads getParticleTempGroup (its two thermostats are "everything but the Drude motion" and
    // "the Drude motion", Ref :426-546), so an array handed over in that mode is ignored -- used as it came, its indices sent the
    // kinetic energy of groups 1.. into the unused and the Drude bins of the three-thermostat block (found by tools/fuzz_soak.py --modes)
    if (tg && d->group) c->group.assign(d->group, d->group + N); else c->group.assign(N, 0);
    if (d->resid) c->resid.assign(d->resid, d->resid + N); else c->resid.clear();

    // pair membership; normalParticles = ascending indices in no pair (Ref :113-137, Cu :111-151)
    std::vector<int> role(N, (int)ROLE_NORMAL), partner(N, -1);
    for (int i = 0; i < P; i++) {
        const int p = c->pair_drude[i], p1 = c->pair_parent[i];
        if (p < 0 || p >= N || p1 < 0 || p1 >= N || p == p1) return fail(TGNH_ERR_ARG, "Drude pair index out of range");
        if (partner[p] != -1 || partner[p1] != -1)
            return fail(TGNH_ERR_UNSUPPORTED, "a particle belongs to more than one Drude pair");
        if (c->mass[p] == 0.0 || c->mass[p1] == 0.0)
            return fail(TGNH_ERR_UNSUPPORTED, "massless Drude particle or parent (the reference's pair arithmetic divides by it)");
        role[p] = ROLE_DRUDE; role[p1] = ROLE_PARENT;
        partner[p] = p1; partner[p1] = p;
        if (tg && c->group[p] != c->group[p1])                                // Cu :145-146
            return fail(TGNH_ERR_GROUP_MISMATCH, "Temperature group for drude particle must be the same as the parent particle");
    }
    c->normal.clear();
    for (int i = 0; i < N; i++) if (role[i] == (int)ROLE_NORMAL) c->normal.push_back(i);
    if (tg) {
        for (int i = 0; i < N; i++)
            if (c->group[i] < 0 || c->group[i] >= d->num_groups) return fail(TGNH_ERR_ARG, "temperature group index out of range");
        if (d->num_groups > MAX_GROUPS) {                                     // K :138-200 sizes its bins by G + 2, no limit: the gather path
            c->generic = true; c->generic_reason = "more than 32 temperature groups";
            if (d->num_groups + 2 > GATHER_MAX_NT)
                return fail(TGNH_ERR_UNSUPPORTED, "more than " + std::to_string(GATHER_MAX_NT - 2) + " temperature groups (a wavefront's kinetic-energy bins no longer fit the LDS)");
        }
        for (int i = 0; i < d->num_constraints; i++) {                        // Cu :186-193
            if (!d->constraint_i) break;                                       // (no arrays: every constraint counts against group 0, local_dof_terms)
            const int a = d->constraint_i[i];
            if (a < 0 || a >= N) return fail(TGNH_ERR_ARG, "constraint index out of range");       // (read again by local_dof_terms)
            if (!d->constraint_j) continue;
            const int b = d->constraint_j[i];
            if (b < 0 || b >= N) return fail(TGNH_ERR_ARG, "constraint index out of range");
            if (c->group[a] != c->group[b])
                return fail(TGNH_ERR_GROUP_MISMATCH, "Temperature group of constrained particles must be the same");
        }
    }

    // residue table (count, first) as the reference builds it (Cu :87-89, :119-125)
    const int R = tg ? d->num_residues : 0;
    c->res_count.assign(R, 0);
    c->res_first.assign(R, -1);
    std::vector<int> res_order;        // residues in order of first appearance (internal index)
    std::vector<int> res_internal(R, -1);
    if (tg) {
        if ((int)c->resid.size() != N) return fail(TGNH_ERR_ARG, "TGNH mode needs resid[N]");
        int prev = -1;
        for (int i = 0; i < N; i++) {
            const int r = c->resid[i];
            if (r < 0 || r >= R) return fail(TGNH_ERR_ARG, "residue index out of range");
            c->res_count[r] += 1;
            if (prev != r) {
                // A residue in several runs (e.g. all Drude particles appended behind the atoms): the reference's table still says
                // (count, first) with `first` the start of the LAST run (Cu :121-124) and its COM kernel walks `count` particles
                // from there (K :90-91), whoever they belong to.  The tiles need molecules in one piece; the gather path reproduces
                // that walk as it is (so does the oracle).
                if (com && c->res_first[r] != -1 && !c->generic) { c->generic = true; c->generic_reason = "particles of a residue are not contiguous"; }
                c->res_first[r] = i;
                if (res_internal[r] == -1) { res_internal[r] = (int)res_order.size(); res_order.push_back(r); }
                prev = r;
            }
        }
        if (com) {      // a molecule of massless sites only: the reference forms v_com = 0 * RECIP(0) = NaN for it (K :86-104) and
                        // every thermostat follows; refuse it rather than reproduce that
            std::vector<double> rmass(R, 0.0);
            for (int i = 0; i < N; i++) rmass[c->resid[i]] += c->mass[i];
            for (int r : res_order)
                if (!(rmass[r] > 0.0)) return fail(TGNH_ERR_UNSUPPORTED, "a molecule has no massive particle (its centre-of-mass velocity is undefined)");
        }
    }

    // Molecules longer than a tile ("big": proteins, polymers) cannot have their COM formed in LDS; theirs comes
    // from a table filled by big_com_kernel, and tiles may cut them anywhere (except through a Drude pair).
    std::vector<char> is_big(R, 0);
    c->big_first.clear(); c->big_count.clear();
    std::vector<int> big_index(R, -1);
    if (com) {
        for (int r : res_order) {
            if (c->res_count[r] > TILE_SLOTS) {
                is_big[r] = 1;
                big_index[r] = (int)c->big_first.size();
                c->big_first.push_back(c->res_first[r]);
                c->big_count.push_back(c->res_count[r]);
            }
        }
    }
    // allowed tile cuts: never through a pair, never through a small molecule when the COM is needed
    std::vector<int> forbid(N + 2, 0);
    for (int i = 0; i < P; i++) {
        const int lo = std::min(c->pair_drude[i], c->pair_parent[i]), hi = std::max(c->pair_drude[i], c->pair_parent[i]);
        forbid[lo + 1] += 1; forbid[hi + 1] -= 1;
    }
    if (com) {
        for (int r : res_order) {
            if (is_big[r]) continue;
            // (a residue in several runs -- the gather path, decided above -- has its `count` particles counted from the start of its
            // LAST run: that walk may leave the array; found by tests/test_desc_fuzz.py as a write behind `forbid`)
            const int lo = c->res_first[r], hi = std::min(lo + c->res_count[r] - 1, N - 1);
            forbid[lo + 1] += 1; forbid[hi + 1] -= 1;
        }
    }
    for (int i = 1; i <= N + 1; i++) forbid[i] += forbid[i - 1];
    std::vector<int> res_starts_before(N + 1, 0);      // # residues whose first slot < i
    if (com) {
        std::vector<char> is_start(N, 0);
        for (int r : res_order) is_start[c->res_first[r]] = 1;
        for (int i = 0; i < N; i++) res_starts_before[i + 1] = res_starts_before[i] + is_start[i];
    }
    int align = 1;
#ifdef TGNH_TUNING
    if (const char* e = getenv("TGNH_TILE_ALIGN")) { align = atoi(e); if (align < 1) align = 1; }
#endif
    c->tile_start.clear(); c->tile_res.clear();
    // residues overlapping [start, e): those starting inside, plus one that started before `start`
    auto entries_in = [&](int start, int e) {
        int n = res_starts_before[e] - res_starts_before[start];
        if (start > 0 && start < N && c->resid[start] == c->resid[start - 1]) n += 1;
        return n;
    };
    // Where the molecular COM is not needed (dualNH; TGNH without the COM group) a tile may end inside a molecule -- but a box of
    // one small molecule then has tiles that start at every phase of it, i.e. as many index-word patterns as the molecule has
    // slots, and a kernel whose lane forms its word again for every tile (dualNH/mixed/resident 200 us per launch where TGNH's
    // 60-slot tiles of whole waters take 186).  So a cut that may go anywhere still prefers a molecule's end when one lies within
    // the last tenth of the tile.
    const bool have_resid = (int)c->resid.size() == N;
    auto mol_cut = [&](int st, int end, int span, auto&& legal) {
        if (com || !have_resid || end >= N) return end;
        for (int e = end; e > st && e >= end - span / 10; e--)
            if (c->resid[e] != c->resid[e - 1] && legal(e)) return e;
        return end;
    };
    int cap = TILE_SLOTS;
#ifdef TGNH_TUNING
    if (const char* e = getenv("TGNH_TILE_CAP")) { int v = atoi(e); if (v >= 64 && v <= TILE_SLOTS) cap = v; }
#endif
    int start = 0;
    while (start < N && !c->generic) {
        int end = std::min(start + cap, N);
        auto ok = [&](int e) {
            if (e < N && forbid[e] > 0) return false;
            if (com && entries_in(start, e) > TILE_RES) return false;
            return true;
        };
        while (end > start && !ok(end)) end--;
        if (end == start) {              // no legal cut within a tile's reach: a Drude far from its parent (K :171-186 gathers by arbitrary
            c->generic = true;           // index), or pairs overlapping so densely that no cut between two of them exists -> the gather path
            c->generic_reason = "a Drude pair (or a chain of overlapping pairs) spans more than one 512-slot tile";
            break;
        }
        end = mol_cut(start, end, cap, ok);
        if (align > 1 && end < N) {           // prefer a cut on an `align`-slot boundary close by
            for (int e = end; e > start && e > end - 64; e--)
                if (e % align == 0 && ok(e)) { end = e; break; }
        }
        c->tile_start.push_back(start);
        start = end;
    }
    c->tile_start.push_back(N);
    c->num_tiles = (int)c->tile_start.size() - 1;
    if (c->generic) return build_gather_topology(c, role, partner, res_order);

    // per-tile residue entries (count, first slot) -- count < 0: big molecule, COM at table index -count-1 --
    // and the packed per-slot words
    std::vector<int2> entries;
    c->tile_res.assign(c->num_tiles + 1, 0);
    c->meta.assign(N, 0);
    for (int t = 0; t < c->num_tiles; t++) {
        c->tile_res[t] = (int)entries.size();
        int prev_res = -1, local = -1;
        for (int i = c->tile_start[t]; i < c->tile_start[t + 1]; i++) {
            if (com && c->resid[i] != prev_res) {
                prev_res = c->resid[i];
                local++;
                entries.push_back(is_big[prev_res] ? make_int2(-(big_index[prev_res] + 1), 0)
                                                   : make_int2(c->res_count[prev_res], c->res_first[prev_res]));
            }
            int off = 0;
            if (partner[i] >= 0) {
                off = partner[i] - i;
                if (partner[i] < c->tile_start[t] || partner[i] >= c->tile_start[t + 1] || off < -1024 || off > 1023)
                    return fail(TGNH_ERR_STATE, "internal: Drude partner outside its tile");
            }
            c->meta[i] = pack_meta((uint32_t)role[i], (uint32_t)c->group[i], off, (uint32_t)(com ? local : 0));
        }
        if (com && local + 1 > TILE_RES) return fail(TGNH_ERR_STATE, "internal: too many molecules in a tile");
    }
    c->tile_res[c->num_tiles] = (int)entries.size();
    if (entries.empty()) entries.push_back(make_int2(0, 0));
    c->res_entries = entries;
    c->num_big = (int)c->big_first.size();

    // Wave tiles for the kinetic-energy passes (wke_kernel, tgnh_internal.h): <= 64 consecutive slots, cut where the
    // 512-slot tiles may be cut (never through a pair, never through a molecule when its COM is needed).  Not possible --
    // a molecule or a pair longer than a wavefront -- leaves the list empty and the KE passes on the tile kernel.
    c->wave_tile.clear(); c->wmeta.clear(); c->num_wtiles = 0;
    if (c->num_big == 0) {
        std::vector<int2> wt;
        bool fits = true;
        for (int st = 0; st < N && fits;) {
            int end = std::min(st + WAVE_SLOTS, N);
            while (end > st && end < N && forbid[end] > 0) end--;
            if (end == st) { fits = false; break; }
            end = mol_cut(st, end, WAVE_SLOTS, [&](int e) { return forbid[e] == 0; });
            int maxn = 1;
            if (com) for (int i = st; i < end; i++) maxn = std::max(maxn, c->res_count[c->resid[i]]);
            wt.push_back(make_int2(st, maxn));
            st = end;
        }
        if (fits) {
            c->num_wtiles = (int)wt.size();
            wt.push_back(make_int2(N, 0));
            c->wmeta.assign(N, 0);
            for (int i = 0; i < N; i++) {
                int pos = 0, n = 1;
                if (com) { const int r = c->resid[i]; pos = i - c->res_first[r]; n = c->res_count[r]; }
                const int off = partner[i] >= 0 ? partner[i] - i : 0;    // inside the wave tile: no cut goes through a pair
                c->wmeta[i] = pack_wmeta((uint32_t)role[i], (uint32_t)c->group[i], off, (uint32_t)pos, (uint32_t)(n - 1));
            }
            c->wave_tile = wt;
        }
    }
    // tiles of identical molecules (PATTERN_WORDS, tgnh_internal.h)
    {
        std::map<std::vector<uint32_t>, uint32_t> ids, wids;
        auto intern = [&](std::map<std::vector<uint32_t>, uint32_t>& m, std::vector<uint32_t>& store, const uint32_t* w, int P,
                          uint32_t limit, uint32_t* id) {
            std::vector<uint32_t> key(w, w + P);
            auto it = m.find(key);
            if (it == m.end()) {
                if (m.size() >= limit) return false;
                it = m.emplace(key, (uint32_t)m.size()).first;
                key.resize(PATTERN_WORDS, 0u);
                store.insert(store.end(), key.begin(), key.end());
            }
            *id = it->second;
            return true;
        };
        c->tile_pat.assign(std::max(c->num_tiles, 1), 0u); c->pattern.clear();
        for (int t = 0; t < c->num_tiles; t++) {
            const int ts = c->tile_start[t], n = c->tile_start[t + 1] - ts;
            // the smallest period that reproduces the tile (a wrong one fails within a few slots); the molecules it spans are
            // read off the molecule index of the slot one period in (a cation and its anion: 2; ten waters, one tagged: 10)
            for (int P = 1; P <= PATTERN_WORDS && P < n; P++) {
                const uint32_t mols = com ? (c->meta[ts + P] >> 21) - (c->meta[ts] >> 21) : 0u;
                if (mols > 255u) break;
                bool same = true;
                for (int k = 0; k < n && same; k++)
                    same = c->meta[ts + k] == c->meta[ts + k % P] + (((uint32_t)(k / P) * mols) << 21);
                if (!same) continue;
                uint32_t id;
                if (intern(ids, c->pattern, c->meta.data() + ts, P, 1u << 16, &id))
                    c->tile_pat[t] = (uint32_t)P | (mols << 8) | (id << 16);
                break;
            }
        }
        c->wtile_pat.assign(std::max(c->num_wtiles, 1), 0u); c->wpattern.clear();
        for (int t = 0; t < c->num_wtiles; t++) {
            const int ws = c->wave_tile[t].x, n = c->wave_tile[t + 1].x - ws;
            for (int P = 1; P <= PATTERN_WORDS && P < n; P++) {
                bool same = true;
                for (int k = 0; k < n && same; k++) same = c->wmeta[ws + k] == c->wmeta[ws + k % P];
                if (!same) continue;
                uint32_t id;
                if (intern(wids, c->wpattern, c->wmeta.data() + ws, P, 1u << 16, &id))
                    c->wtile_pat[t] = (uint32_t)P | (id << 8);
                break;
            }
        }
        if (c->pattern.empty()) c->pattern.assign(PATTERN_WORDS, 0u);
        if (c->wpattern.empty()) c->wpattern.assign(PATTERN_WORDS, 0u);
    }
    if (c->host_only) return TGNH_OK;
    // device copies
    HIP_OK(hipMalloc(&c->d_tile_pat, sizeof(uint32_t) * c->tile_pat.size()));
    HIP_OK(hipMemcpy(c->d_tile_pat, c->tile_pat.data(), sizeof(uint32_t) * c->tile_pat.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->d_pattern, sizeof(uint32_t) * c->pattern.size()));
    HIP_OK(hipMemcpy(c->d_pattern, c->pattern.data(), sizeof(uint32_t) * c->pattern.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->d_wpattern, sizeof(uint32_t) * c->wpattern.size()));
    HIP_OK(hipMemcpy(c->d_wpattern, c->wpattern.data(), sizeof(uint32_t) * c->wpattern.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->d_meta, sizeof(uint32_t) * std::max(N, 1)));
    HIP_OK(hipMemcpy(c->d_meta, c->meta.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->d_tile_start, sizeof(int) * c->tile_start.size()));
    HIP_OK(hipMemcpy(c->d_tile_start, c->tile_start.data(), sizeof(int) * c->tile_start.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->d_tile_res, sizeof(int) * c->tile_res.size()));
    HIP_OK(hipMemcpy(c->d_tile_res, c->tile_res.data(), sizeof(int) * c->tile_res.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&c->d_res_table, sizeof(int2) * c->res_entries.size()));
    HIP_OK(hipMemcpy(c->d_res_table, c->res_entries.data(), sizeof(int2) * c->res_entries.size(), hipMemcpyHostToDevice));
    if (!c->wave_tile.empty()) {
        std::vector<int2> packed = c->wave_tile;           // device copy: y = largest molecule | period << 8 | pattern << 16
        for (int t = 0; t < c->num_wtiles; t++) packed[t].y |= (int)(c->wtile_pat[t] << 8);
        HIP_OK(hipMalloc(&c->d_wave_tile, sizeof(int2) * packed.size()));
        HIP_OK(hipMemcpy(c->d_wave_tile, packed.data(), sizeof(int2) * packed.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&c->d_wmeta, sizeof(uint32_t) * N));
        HIP_OK(hipMemcpy(c->d_wmeta, c->wmeta.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice));
    }
    if (c->num_big) {
        std::vector<int2> bt(c->num_big);
        for (int k = 0; k < c->num_big; k++) bt[k] = make_int2(c->big_count[k], c->big_first[k]);
        HIP_OK(hipMalloc(&c->d_big_table, sizeof(int2) * c->num_big));
        HIP_OK(hipMemcpy(c->d_big_table, bt.data(), sizeof(int2) * c->num_big, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&c->d_big_com, 32 * (size_t)c->num_big));           // mixed4 per big molecule
        HIP_OK(hipMemset(c->d_big_com, 0, 32 * (size_t)c->num_big));
    }
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// A2: degrees of freedom (additive local terms) and thermostat block
// ---------------------------------------------------------------------------
static void local_dof_terms(tgnh_context* c) {
    const tgnh_desc& d = c->d;
    const int N = d.num_particles, P = d.num_pairs;
    const int NT = c->L.NT;
    c->local_terms.assign(NT, 0.0);
    if (d.mode == TGNH_MODE_DUALNH) {
        double real = 0;
        for (int i = 0; i < N; i++) real += (c->mass[i] == 0.0 ? 0 : 3);      // Ref :119
        real -= 3.0 * P;                                                      // Ref :133
        real -= d.num_constraints;                                            // Ref :157
        c->local_terms[0] = real;
        c->local_terms[2] = 3.0 * P;                                          // Ref :134
        return;
    }
    const int G = d.num_groups, R = d.num_residues;
    std::vector<double> resInv(R, 0.0);                                       // API :147-153
    {
        std::vector<double> rm(R, 0.0);
        for (int i = 0; i < N; i++) rm[c->resid[i]] += c->mass[i];
        for (int r = 0; r < R; r++) resInv[r] = 1.0 / rm[r];
    }
    std::vector<double> dof(G, 0.0), red(G, 0.0);
    for (int i = 0; i < N; i++) {                                             // Cu :126-133
        if (c->mass[i] != 0.0) {
            dof[c->group[i]] += 3;
            if (d.use_com_temp_group) red[c->group[i]] += 3 * c->mass[i] * resInv[c->resid[i]];
        }
    }
    for (int i = 0; i < P; i++) dof[c->group[c->pair_drude[i]]] -= 3;         // Cu :148
    for (int i = 0; i < d.num_constraints; i++) {                             // Cu :195
        if (d.constraint_i) dof[c->group[d.constraint_i[i]]] -= 1; else dof[0] -= 1;
    }
    for (int g = 0; g < G; g++) c->local_terms[g] = dof[g] - red[g];          // Cu :219
    c->local_terms[G] = d.use_com_temp_group ? 3.0 * R : 0.0;                 // Cu :197-199
    c->local_terms[G + 1] = 3.0 * P;                                          // Cu :149, :201
}

static tgnh_status finalize_thermostat(tgnh_context* c) {
    const tgnh_desc& d = c->d;
    ChainLayout& L = c->L;
    const int NT = L.NT, C = L.C;
    c->dof = c->global_terms;
    if (d.has_cm_motion_remover) {
        if (d.mode == TGNH_MODE_DUALNH) c->dof[0] -= 3;                       // Ref :158-165
        else if (d.use_com_temp_group) c->dof[L.G] -= 3;                      // Cu :204-212
    }
    c->nkbt.assign(NT, 0.0);
    std::vector<double> st(L.total, 0.0);
    const double tau2 = std::pow(d.coupling_time, 2), tauD2 = std::pow(d.drude_coupling_time, 2);
    if (d.mode == TGNH_MODE_DUALNH) {
        const double realNkbT = c->dof[0] * c->realkbT, drudeNkbT = c->dof[2] * c->drudekbT;   // Ref :168-169
        c->nkbt[0] = realNkbT; c->nkbt[2] = drudeNkbT;
        double* etaMass = st.data() + L.off_etaMass;
        double* etaDot = st.data() + L.off_etaDot;
        double* etaDotDot = st.data() + L.off_etaDotDot;
        etaMass[0] = realNkbT * tau2;                                         // Ref :170-171
        etaMass[1] = drudeNkbT * tauD2;
        const int ntg = L.numTempGroup;
        if (L.use_drude_chains) {                                             // Ref :192-205
            for (int ich = 1; ich < C; ich++) {
                etaMass[2 * ich] = c->realkbT * tau2;
                etaMass[2 * ich + 1] = c->drudekbT * tauD2;
                etaDotDot[ich * ntg] = (etaMass[(ich - 1) * ntg] * etaDot[(ich - 1) * ntg] * etaDot[(ich - 1) * ntg] - c->realkbT) / etaMass[ich * ntg];
                etaDotDot[ich * ntg + 1] = (etaMass[(ich - 1) * ntg + 1] * etaDot[(ich - 1) * ntg + 1] * etaDot[(ich - 1) * ntg + 1] - c->drudekbT) / etaMass[ich * ntg + 1];
            }
        } else {                                                              // Ref :206-214
            for (int ich = 1; ich < C; ich++) {
                etaMass[ich + 1] = c->realkbT * tau2;
                etaDotDot[ich * ntg + 1] = (etaMass[(ich - 1) * ntg + 1] * etaDot[(ich - 1) * ntg + 1] * etaDot[(ich - 1) * ntg + 1] - c->realkbT) / etaMass[ich * ntg + 1];
            }
        }
    } else {
        const int G = L.G;
        const double realUnit = c->realkbT * tau2, drudeUnit = c->drudekbT * tauD2;   // Cu :216-217
        for (int i = 0; i < G + 1; i++) {                                     // Cu :218-225
            c->nkbt[i] = c->dof[i] * c->realkbT;
            double* em = st.data() + L.off_etaMass + i * C;
            double* edd = st.data() + L.off_etaDotDot + i * C;
            em[0] = c->dof[i] * realUnit;
            for (int ich = 1; ich < C; ich++) { em[ich] = realUnit; edd[ich] = (em[ich - 1] * 0.0 - c->realkbT) / em[ich]; }
        }
        const int itg = G + 1;                                                // Cu :227-235
        c->nkbt[itg] = c->dof[itg] * c->drudekbT;
        double* em = st.data() + L.off_etaMass + itg * C;
        double* edd = st.data() + L.off_etaDotDot + itg * C;
        em[0] = c->dof[itg] * drudeUnit;
        for (int ich = 1; ich < C; ich++) {
            em[ich] = drudeUnit;
            if (L.use_drude_chains) edd[ich] = (em[ich - 1] * 0.0 - c->drudekbT) / em[ich];
        }
    }
    for (int i = 0; i < NT; i++) {
        st[L.off_nkbt + i] = c->nkbt[i];
        st[L.off_scale + i] = 1.0; st[L.off_scale_a + i] = 1.0; st[L.off_scale_b + i] = 1.0;
    }
    c->h_state = st;
    if (!c->host_only) {
        HIP_OK(hipMemcpy(c->d_state, st.data(), sizeof(double) * L.total, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(c->d_stage, st.data(), sizeof(double) * L.total, hipMemcpyHostToDevice));
    }
    c->chain_pending = false; c->stage_pending = false; c->carry_pending = false; c->ke_carry = false;
    c->scale_pending = false; c->kick_pending = false; c->first_half_done = false; c->end_pending = false;
    return TGNH_OK;
}

static void make_layout(tgnh_context* c) {
    const tgnh_desc& d = c->d;
    ChainLayout& L = c->L;
    L.mode = d.mode;
    L.C = d.num_nh_chains;
    L.use_drude_chains = d.use_drude_nh_chains ? 1 : 0;
    if (d.mode == TGNH_MODE_DUALNH) {
        L.G = 1; L.NT = 3;
        const int C = L.C;
        if (L.use_drude_chains) { L.numTempGroup = 2; L.idxMaxNHChains = 2 * C - 1; L.iNumNHChains = 2 * C; }   // Ref :139-154
        else { L.numTempGroup = 1; L.idxMaxNHChains = C; L.iNumNHChains = C + 1; }
        const int n = L.use_drude_chains ? 2 * C : C + 1;
        L.len_eta = n; L.len_etaDotDot = n; L.len_etaMass = n; L.len_etaDot = n + 2;          // Ref :216-217
        L.c1_shift = 1; L.c1_mul = 1; L.c1_add = 2; L.c1_unused = 1; L.c1_guard_below = 0;
        L.c1_quirk = L.use_drude_chains ? 0 : 1;
    } else {
        L.G = d.num_groups; L.NT = L.G + 2;
        L.len_eta = L.NT * L.C; L.len_etaDotDot = L.NT * L.C; L.len_etaMass = L.NT * L.C;     // Cu :94-97
        L.len_etaDot = L.NT * (L.C + 1);
        L.numTempGroup = L.idxMaxNHChains = L.iNumNHChains = 0;
        L.c1_shift = 0; L.c1_mul = 2; L.c1_add = 1; L.c1_unused = -1; L.c1_guard_below = L.NT - 1;
        L.c1_quirk = 0;
    }
    int o = 0;
    L.off_eta = o; o += L.len_eta;
    L.off_etaDot = o; o += L.len_etaDot;
    L.off_etaDotDot = o; o += L.len_etaDotDot;
    L.off_etaMass = o; o += L.len_etaMass;
    L.off_nkbt = o; o += L.NT;
    L.off_ke = o; o += L.NT;
    o = (o + 1) & ~1;                         // 16-byte aligned: this one is handed to the all-reduce hook
    L.off_ke_red = o; o += L.NT;
    L.off_scale = o; o += L.NT;
    L.off_scale_a = o; o += L.NT;
    L.off_scale_b = o; o += L.NT;
    L.off_kesum = o; o += 1;
    L.off_ke_post = o; o += L.NT;
    L.total = o;
}

// ---------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------
static void exchange_release(tgnh_context* c) {
    for (void* p : c->x_opened) (void)hipIpcCloseMemHandle(p);
    c->x_opened.clear();
    if (c->x_mailbox) (void)hipFree(c->x_mailbox);
    if (c->d_x_seq) (void)hipFree(c->d_x_seq);
    if (c->d_x_dead) (void)hipFree(c->d_x_dead);
    if (c->d_x_peers) (void)hipFree(c->d_x_peers);
    if (c->d_x_stat) (void)hipFree(c->d_x_stat);
    c->d_x_stat = nullptr;
    c->x_mailbox = nullptr; c->d_x_seq = nullptr; c->d_x_dead = nullptr; c->d_x_peers = nullptr;
    c->xchg_on = false;
}

static void free_device(tgnh_context* c) {
    if (c->d_meta) (void)hipFree(c->d_meta);
    if (c->d_tile_start) (void)hipFree(c->d_tile_start);
    if (c->d_tile_res) (void)hipFree(c->d_tile_res);
    if (c->d_res_table) (void)hipFree(c->d_res_table);
    if (c->d_wave_tile) (void)hipFree(c->d_wave_tile);
    if (c->d_wmeta) (void)hipFree(c->d_wmeta);
    if (c->d_tile_pat) (void)hipFree(c->d_tile_pat);
    if (c->d_pattern) (void)hipFree(c->d_pattern);
    if (c->d_wpattern) (void)hipFree(c->d_wpattern);
    if (c->d_sflag) (void)hipFree(c->d_sflag);
    if (c->d_sbase) (void)hipFree(c->d_sbase);
    if (c->d_sites) (void)hipFree(c->d_sites);
    if (c->d_lat_tab) (void)hipFree(c->d_lat_tab);
    if (c->d_big_table) (void)hipFree(c->d_big_table);
    if (c->d_big_com) (void)hipFree(c->d_big_com);
    if (c->d_g_group) (void)hipFree(c->d_g_group);
    if (c->d_g_resid) (void)hipFree(c->d_g_resid);
    if (c->d_g_res_table) (void)hipFree(c->d_g_res_table);
    if (c->d_g_partner) (void)hipFree(c->d_g_partner);
    if (c->d_g_com) (void)hipFree(c->d_g_com);
    if (c->d_g_scratch) (void)hipFree(c->d_g_scratch);
    if (c->d_g_x0) (void)hipFree(c->d_g_x0);
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_stage) (void)hipFree(c->d_stage);
    exchange_release(c);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status_seen) (void)hipHostFree(c->h_status_seen);
    c->h_status_seen = nullptr;
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->d_sync) (void)hipFree(c->d_sync);
    if (c->d_rows) (void)hipFree(c->d_rows);
    if (c->self_box) (void)hipFree(c->self_box);
    if (c->d_self_misc) (void)hipFree(c->d_self_misc);
    if (c->d_cl_atoms) (void)hipFree(c->d_cl_atoms);
    if (c->d_cl_dist) (void)hipFree(c->d_cl_dist);
    if (c->d_vs_atoms) (void)hipFree(c->d_vs_atoms);
    if (c->d_vs_w) (void)hipFree(c->d_vs_w);
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    c->ev_pool.clear();
}

extern "C" tgnh_status tgnh_create(const tgnh_desc* d, tgnh_handle* out) {
    if (!d || !out) return fail(TGNH_ERR_ARG, "null argument");
    if (d->struct_size != sizeof(tgnh_desc)) return fail(TGNH_ERR_ARG, "tgnh_desc size mismatch (ABI)");
    if (d->mode != TGNH_MODE_DUALNH && d->mode != TGNH_MODE_TGNH) return fail(TGNH_ERR_ARG, "bad mode");
    if (d->precision < TGNH_PREC_SINGLE || d->precision > TGNH_PREC_DOUBLE) return fail(TGNH_ERR_ARG, "bad precision");
    if (d->num_particles < 1 || d->num_pairs < 0 || !d->mass || (d->num_pairs && (!d->pair_drude || !d->pair_parent)))
        return fail(TGNH_ERR_ARG, "bad particle / pair arrays");
    if (d->padded_num_particles < d->num_particles) return fail(TGNH_ERR_ARG, "padded_num_particles < num_particles");
    if (d->num_constraints < 0) return fail(TGNH_ERR_ARG, "negative num_constraints");
    if (3LL * d->padded_num_particles > 2147483647LL)      // force[i + 2 paddedN] in 32-bit indices, here as in the reference's kernels (K :318-320)
        return fail(TGNH_ERR_UNSUPPORTED, "more than 715 827 882 padded particle slots: the index into the third force plane leaves 32 bits");
    if (d->num_nh_chains < 1 || d->drude_steps_per_real_step < 1) return fail(TGNH_ERR_ARG, "numNHChains and drudeStepsPerRealStep must be >= 1");
    if (d->mode == TGNH_MODE_TGNH && (d->num_groups < 1 || d->num_residues < 1 || !d->group || !d->resid))
        return fail(TGNH_ERR_ARG, "TGNH mode needs temperature groups and residues");
    if (!(d->max_drude_distance >= 0)) return fail(TGNH_ERR_ARG, "setMaxDrudeDistance: Distance cannot be negative");   // API :98-99 (NaN neither)
    if (d->flags & ~(uint32_t)(TGNH_FLAG_DEFER_SCALE | TGNH_FLAG_RESIDENT_STEP | TGNH_FLAG_WAVE_TILES | TGNH_FLAG_TRUST_STATE_CHANGED | TGNH_FLAG_GATHER))
        return fail(TGNH_ERR_ARG, "unknown bits in tgnh_desc.flags (a newer header than this library?)");
    if (d->mode == TGNH_MODE_DUALNH && d->num_pairs == 0)   // Ref :181 reads pairParticles[0]; its chain divides by the Drude thermostat mass 0
        return fail(TGNH_ERR_UNSUPPORTED, "dualNH mode needs at least one Drude pair (the Reference platform does too)");
    if (!(d->step_size > 0) || !std::isfinite(d->step_size)) return fail(TGNH_ERR_ARG, "step size must be positive");
    bool long_chain = false;
    {   // the chain kernel keeps chains longer than 4 links (16 in TGNH mode: chain_long_kernel) in a 2048-double LDS scratch; what
        // does not fit runs a thermostat per thread with its links in global memory (gather_chain_kernel, TGNH mode)
        const long need = d->mode == TGNH_MODE_TGNH ? (long)(d->num_groups + 2) * (4L * d->num_nh_chains + 1)
                                                    : 4L * (2L * d->num_nh_chains + 4);
        if (d->num_nh_chains > (d->mode == TGNH_MODE_TGNH ? 16 : 4) && need > 2048) {
            if (d->mode != TGNH_MODE_TGNH) return fail(TGNH_ERR_UNSUPPORTED, "numNHChains too large for the on-device chain");
            long_chain = true;
        }
    }
    // device == -1: host-only handle for the host logic (topology, tiles, dof); every launch on it fails
    const bool host_only = d->device == -1;
    if (!host_only) {
        int ndev = 0;
        HIP_OK(hipGetDeviceCount(&ndev));
        if (d->device < 0 || d->device >= ndev) return fail(TGNH_ERR_HIP, "no such HIP device (the HIP path needs an MI355X; there is no CPU fallback)");
        HIP_OK(hipSetDevice(d->device));
    }

    tgnh_context* c = new tgnh_context();
    c->host_only = host_only;
    c->d = *d;
    c->device = d->device;
    c->realkbT = d->kB * d->temperature;                                      // Ref :107-108, Cu :80-81
    c->drudekbT = d->kB * d->drude_temperature;
    make_layout(c);
    c->gb = c->L.G <= 1 ? 1 : (c->L.G <= 4 ? 4 : (c->L.G <= 8 ? 8 : 0));   // 0: KE bins in LDS
    if (long_chain) { c->generic = true; c->generic_reason = "a chain too long for the LDS-resident form"; }
    else if (d->flags & TGNH_FLAG_GATHER) { c->generic = true; c->generic_reason = "asked for (TGNH_FLAG_GATHER)"; }
    if (!host_only) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    }
    tgnh_status rc = build_topology(c, d);
    if (rc != TGNH_OK) { free_device(c); delete c; return rc; }
    if (c->generic) {
        // The gather path steps in the reference's own pass structure, velocities never lagging: the flags that change the
        // structure are dropped (all of them leave the trajectory what it is; tgnh_flush has nothing to do, state setters
        // between steps are allowed as without TGNH_FLAG_DEFER_SCALE)
        c->d.flags &= ~(TGNH_FLAG_DEFER_SCALE | TGNH_FLAG_RESIDENT_STEP | TGNH_FLAG_TRUST_STATE_CHANGED | TGNH_FLAG_WAVE_TILES);
        c->gather_chain = c->d.mode == TGNH_MODE_TGNH && (c->L.NT > MAX_GROUPS + 2 || long_chain);
    }
    // KE passes and the one-launch step over wave tiles: register bins (G <= 8), and tiles that fill their wavefront -- a wave
    // tile ends where a molecule does, so 35-slot cations leave 45 of 64 lanes busy and the 512-slot tiles, cut the same way but
    // eight times as long, win (ionic liquid 100 k: 42.0 k steps/s on the tile kernels, 40.3 k on wave tiles; 60-slot water
    // tiles: 94 % full)
    c->wave_ke = !c->generic && !c->wave_tile.empty() && c->gb != 0 &&
                 ((d->flags & TGNH_FLAG_WAVE_TILES) || (double)d->num_particles >= 0.9 * WAVE_SLOTS * (double)c->num_wtiles);
#ifdef TGNH_TUNING
    if (const char* e = getenv("TGNH_WAVE_KE")) c->wave_ke = c->wave_ke && e[0] != '0';
#endif
    {   // s^2 KE is the exact post-rescale KE only if no molecule spans two temperature groups: v_rel = v - v_com of such a
        // molecule is scaled by two different factors, which moves its centre of mass (K :260-300)
        bool inside = true;
        if (d->mode == TGNH_MODE_TGNH && d->use_com_temp_group) {
            std::vector<int> g0(d->num_residues, -1);      // (by particle, not by (first, count): a residue may come in several runs)
            for (int i = 0; i < d->num_particles && inside; i++) {
                if (c->mass[i] == 0.0) continue;
                int& g = g0[c->resid[i]];
                if (g == -1) g = c->group[i];
                else if (g != c->group[i]) inside = false;
            }
        }
        if ((c->d.flags & TGNH_FLAG_DEFER_SCALE) && !inside) { free_device(c); delete c; return fail(TGNH_ERR_UNSUPPORTED, "DEFER_SCALE needs every molecule inside one temperature group"); }
        // TRUST_STATE_CHANGED (the reference's pass structure without the begin half's KE pass) asks the same of the topology;
        // where it does not hold the flag is ignored -- the handle recomputes, as without it (tgnh_get_pending_state bit 9 never shows)
        c->carry_ok = (c->d.flags & TGNH_FLAG_TRUST_STATE_CHANGED) && !(c->d.flags & TGNH_FLAG_DEFER_SCALE) && inside;
    }
    local_dof_terms(c);
    c->global_terms = c->local_terms;
    // constraint arrays are only needed during create
    c->d.mass = nullptr; c->d.pair_drude = c->d.pair_parent = c->d.group = c->d.resid = nullptr;
    c->d.constraint_i = c->d.constraint_j = nullptr;

    c->grid = GRID_CAP;                       // partials are sized for the largest grid
#ifdef TGNH_TUNING       // environment knobs exist in tuning builds only (tools/build_variant.py -DTGNH_TUNING)
    if (const char* e = getenv("TGNH_GRID")) { int g = atoi(e); if (g >= 1) c->grid_override = std::min(g, GRID_CAP); }
#endif
    {   // One-link chains run inside the rescale launch: one wavefront per work-group computes the factors while
        // the other three have their tile loads in flight, so the chain (~3.5 us) costs the launch nothing, and
        // the chain launch that remains only sums the partial rows (profiles/r01_tuning_sweep.log: +7 % steps/s
        // at 625 k slots, +1.5 % at 5 M).
        bool want = true;
        // The partial rows are summed in that prologue too (no sum launch) up to 2 M slots: beyond, the launches are
        // bandwidth-bound, the gain shrinks to 0.7 % and the row read would only lengthen the dominant launch
        c->inline_sum_all = d->num_particles < 2000000;
#ifdef TGNH_TUNING
        if (const char* e = getenv("TGNH_INLINE_CHAIN")) want = e[0] != '0';
        if (const char* e4 = getenv("TGNH_INLINE_SUM_ROWS")) c->inline_sum_rows = atoi(e4);   // 0 = never
        if (const char* e5 = getenv("TGNH_INLINE_SUM_ALL")) c->inline_sum_all = e5[0] != '0';
        if (const char* e3 = getenv("TGNH_ALTERNATE_SWEEPS")) c->alternate_sweeps = e3[0] != '0';
#endif
        // dualNH qualifies too: with useDrudeNHChains its real and Drude chains are independent (Chain1Map), without
        // them coupled through one shuffle per sub-step (chain1q_run)
        // Chains of 2-4 links too, but in instantiations of their own that hold two work-groups per compute unit where the
        // one-link kernels hold three (the links' registers): taken below 2.5 M slots.  Beyond, the streaming launches are
        // bandwidth-bound and keep their occupancy -- chain_kernel's 18 us cost less.  (The limit was 1 M slots while a launch's
        // chain wavefront ran the real thermostats and the Drude thermostat one after the other; with both in one pass,
        // chain_both_fast, three links inside the launches read +16 % at 625 k slots, +7 % at 1.25 M, +3..6 % at 2 M and
        // -4 % / +1 % (one launch per step / deferred) at 5 M: tools/micro/inline_multi_threshold.py,
        // profiles/r04_inline_multi_threshold.txt)
        int inline_multi_max = 2500000;
#ifdef TGNH_TUNING
        if (const char* e6 = getenv("TGNH_INLINE_MULTI_MAX")) inline_multi_max = atoi(e6);
#endif
        c->inline_chain = want && !c->generic && (c->L.C == 1 || (c->L.C <= 4 && d->num_particles < inline_multi_max));
        if (c->L.total > 256 && c->L.C > 1) c->inline_chain = c->inline_chain && false;     // (wstep_kernel parks the block in 256 doubles)
    }
    auto alloc = [&]() -> tgnh_status {
        if (host_only) return TGNH_OK;
        HIP_OK(hipMalloc(&c->d_partials, sizeof(double) * (size_t)(c->grid + c->num_big) * c->L.NT));
        HIP_OK(hipMemset(c->d_partials, 0, sizeof(double) * (size_t)(c->grid + c->num_big) * c->L.NT));
        HIP_OK(hipMalloc(&c->d_state, sizeof(double) * c->L.total));
        HIP_OK(hipMalloc(&c->d_stage, sizeof(double) * c->L.total));
        HIP_OK(hipMalloc(&c->d_status, sizeof(uint32_t)));
        HIP_OK(hipMemset(c->d_status, 0, sizeof(uint32_t)));
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&c->h_status_seen), sizeof(uint32_t), hipHostMallocDefault));
        *c->h_status_seen = 0;

        if (c->gather_chain && c->L.C > 4) HIP_OK(hipMalloc(&c->d_g_scratch, sizeof(double) * (size_t)c->L.NT * (4 * c->L.C + 1)));
        HIP_OK(hipMalloc(&c->d_scalar, sizeof(double) * (1 + PLAIN_KE_PARTS)));
        HIP_OK(hipMalloc(&c->d_sync, 4 * sizeof(unsigned int)));
        HIP_OK(hipMemset(c->d_sync, 0, 4 * sizeof(unsigned int)));
        if (c->wave_ke && !(c->d.flags & TGNH_FLAG_RESIDENT_STEP)) {      // the tagged rows of wke_kernel's tail sum
            const size_t rb = sizeof(unsigned long long) * 2 * (size_t)GRID_CAP * CHAIN_INLINE_SUM_NT;
            void* p = nullptr;
            HIP_OK(hipExtMallocWithFlags(&p, rb, TGNH_MEETING_MEM));
            c->d_rows = static_cast<unsigned long long*>(p);
            HIP_OK(hipMemset(c->d_rows, 0, rb));
        }
        if (c->d.flags & TGNH_FLAG_RESIDENT_STEP) {
            // step_kernel's meeting place: the work-groups' tagged rows, and a private one-rank mailbox that carries the
            // sums from work-group 0 to all the others when no sharded exchange is attached.  Both stay on this device:
            // fine-grained memory (a flag stored by one work-group is seen by a polling one on another XCD after 0.37-0.39 us,
            // uncached memory takes 0.58-0.63: tools/micro/hop_probe.hip); the mailboxes peers store into are uncached
            const size_t rb = sizeof(unsigned long long) * 2 * (size_t)GRID_CAP * CHAIN_INLINE_SUM_NT;
            void* p = nullptr;
            HIP_OK(hipExtMallocWithFlags(&p, rb, TGNH_MEETING_MEM));
            c->d_rows = static_cast<unsigned long long*>(p);
            HIP_OK(hipMemset(c->d_rows, 0, rb));
            HIP_OK(hipExtMallocWithFlags(&p, XCHG_MAILBOX_BYTES(1), TGNH_MEETING_MEM));
            c->self_box = static_cast<unsigned long long*>(p);
            HIP_OK(hipMemset(c->self_box, 0, XCHG_MAILBOX_BYTES(1)));
            HIP_OK(hipMalloc(&c->d_self_misc, 4 * sizeof(unsigned long long)));     // [0] seq, [1] dead latch, [2] peers[0]
            HIP_OK(hipMemset(c->d_self_misc, 0, 4 * sizeof(unsigned long long)));
            HIP_OK(hipMemcpy(c->d_self_misc + 2, &c->self_box, sizeof(unsigned long long*), hipMemcpyHostToDevice));
            c->self_x = XchgArgs{};
            c->self_x.on = 1; c->self_x.world = 1; c->self_x.rank = 0;
            c->self_x.peers = reinterpret_cast<unsigned long long* const*>(c->d_self_misc + 2);
            c->self_x.mine = c->self_box;
            c->self_x.seq = c->d_self_misc;
            c->self_x.dead = reinterpret_cast<unsigned int*>(c->d_self_misc + 1);
            c->self_x.status = c->d_status;
            // How many work-groups of step_kernel per compute unit are resident TOGETHER?  The occupancy API's answer is
            // checked by a census launch (every work-group checks in and waits for all the others, bounded); one fewer per
            // unit is tried until a grid passes.  0 = none did: the handle steps the DEFER_SCALE way.
            if (c->gb != 0 && c->inline_chain && c->L.NT <= CHAIN_INLINE_SUM_NT) {
                // (the kind with the largest footprint this handle will launch: a whole deferred step, or the plain begin half)
                const int kind = (c->d.flags & TGNH_FLAG_DEFER_SCALE) ? 0 : 1;
                const size_t lds = tile_lds_bytes(c->d.precision, step_kind_ops2(kind), true, true);
                // (step_kernel runs one-link chains only; longer ones have wstep_kernel below, or the launches)
                for (int per_cu = c->L.C == 1 ? std::min(step_blocks_per_cu(c->d.precision, c->gb, kind, lds), 8) : 0; per_cu >= 1 && !c->resident_per_cu; per_cu--) {
                    TileArgs a{};
                    a.census = 1; a.sync = c->d_sync;
                    HIP_OK(hipMemset(c->d_sync + 2, 0, 2 * sizeof(unsigned int)));
                    const int grid = std::min(per_cu * c->num_cus, GRID_CAP);
                    HIP_OK(launch_step(c->d.precision, c->gb, kind, a, grid, lds, (hipStream_t)0));
                    unsigned int res[2] = {0, 1};
                    HIP_OK(hipMemcpy(res, c->d_sync + 2, sizeof(res), hipMemcpyDeviceToHost));
                    if (res[0] == (unsigned)grid && res[1] == 0) c->resident_per_cu = per_cu;
                }
                // the same count for wstep_kernel, which runs the whole deferred step when the topology has wave tiles
                const bool multi = c->L.C > 1;
                bool want_w = c->wave_ke && (c->d.flags & TGNH_FLAG_DEFER_SCALE) && (c->resident_per_cu > 0 || multi);
                // dualNH's coupled chain (useDrudeNHChains = false) of 2-4 links: wstep_kernel has no room for its fast form beside its
                // 228 registers (chainN_run<false>), the rescale launches have -- and every kernel of every rank must run the SAME
                // arithmetic (replicated chains stay bit-identical): such a handle steps the DEFER_SCALE way, chain inside the launch
                if (c->d.mode == TGNH_MODE_DUALNH && !c->L.use_drude_chains && multi) want_w = false;
#ifdef TGNH_TUNING
                if (const char* e = getenv("TGNH_WSTEP")) want_w = want_w && e[0] != '0';
#endif
                for (int per_cu = want_w ? std::min(wstep_blocks_per_cu(c->d.precision, c->gb, multi), 8) : 0; per_cu >= 1 && !c->wresident_per_cu; per_cu--) {
                    TileArgs a{};
                    a.census = 1; a.sync = c->d_sync;
                    HIP_OK(hipMemset(c->d_sync + 2, 0, 2 * sizeof(unsigned int)));
                    const int grid = std::min(per_cu * c->num_cus, GRID_CAP);
                    HIP_OK(launch_wstep(c->d.precision, c->gb, multi, a, grid, (hipStream_t)0));
                    unsigned int res[2] = {0, 1};
                    HIP_OK(hipMemcpy(res, c->d_sync + 2, sizeof(res), hipMemcpyDeviceToHost));
                    if (res[0] == (unsigned)grid && res[1] == 0) c->wresident_per_cu = per_cu;
                }
            }
        }
        return TGNH_OK;
    };
    rc = alloc();
    if (rc == TGNH_OK) rc = finalize_thermostat(c);
    if (rc != TGNH_OK) { free_device(c); delete c; return rc; }
    *out = c;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_destroy(tgnh_handle h) {
    CHECK_H(h);
    if (h->rccl_comm) (void)tgnh_rccl_shutdown(h);
    if (!h->host_only) { (void)hipSetDevice(h->device); free_device(h); }
    delete h;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_bind_buffers(tgnh_handle h, void* posq, void* posq_correction, void* velm,
                                         const void* force, void* pos_delta) {
    CHECK_H(h);
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle (device -1): no GPU work can be launched on it");
    if (!posq || !velm || !force) return fail(TGNH_ERR_ARG, "posq, velm and force are required");
    if (h->d.precision == TGNH_PREC_MIXED && !posq_correction) return fail(TGNH_ERR_ARG, "mixed precision needs posqCorrection");
    if ((h->kick_pending || h->end_pending) && (velm != h->velm || force != h->force))
        return fail(TGNH_ERR_STATE, "tgnh_bind_buffers: velm / force may not be rebound while a deferred half kick is pending (tgnh_flush first)");
    if (posq == h->posq && posq_correction == h->posq_corr && velm == h->velm && force == h->force && pos_delta == h->pos_delta)
        return TGNH_OK;                                   // (the glue binds at every step: the same arrays, nothing to do)
    {   // Every kernel indexes these arrays by slot without a bound of its own: an allocation shorter than N slots (or 3 planes
        // of `padded`) would be a write off its end on the device.  Where the runtime knows the allocation a pointer lies in,
        // what is left of it behind the pointer must cover what the launches touch (a sub-allocation of a caching allocator
        // passes with its block's size: a lower bound, but the gross cases -- a float4 array bound as double4, N for padded --
        // are caught here, on the host, with a message).
        HIP_OK(hipSetDevice(h->device));
        const size_t N = (size_t)h->d.num_particles, P = (size_t)h->d.padded_num_particles;
        const size_t r4 = h->d.precision == TGNH_PREC_DOUBLE ? 32 : 16, m4 = h->d.precision == TGNH_PREC_SINGLE ? 16 : 32;
        struct { const void* p; size_t need; const char* name; } bufs[] = {
            {posq, N * r4, "posq"}, {posq_correction, N * 16, "posqCorrection"}, {velm, N * m4, "velm"},
            {force, 3 * P * sizeof(long long), "force"}, {pos_delta, N * m4, "posDelta"}};
        for (const auto& b : bufs) {
            if (!b.p) continue;
            hipDeviceptr_t base = nullptr; size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, const_cast<void*>(b.p)) != hipSuccess) { (void)hipGetLastError(); continue; }   // not known to the runtime (mapped by other means): the caller's word is taken
            const size_t left = size - (size_t)(static_cast<const char*>(b.p) - static_cast<const char*>(base));
            if (left < b.need)
                return fail(TGNH_ERR_ARG, std::string("tgnh_bind_buffers: ") + b.name + " has " + std::to_string(left) + " bytes behind the pointer, the launches touch " + std::to_string(b.need));
        }
    }
    h->ke_carry = false;                                  // (other buffers: nothing computed from the old ones carries over)
    h->posq = posq; h->posq_corr = posq_correction; h->velm = velm; h->force = force; h->pos_delta = pos_delta;
    return TGNH_OK;
}

static tgnh_status deferred_guard(tgnh_handle h, const char* what) {
    if (h->first_half_done || h->end_pending)
        return fail(TGNH_ERR_STATE, std::string(what) + ": not allowed between steps with TGNH_FLAG_DEFER_SCALE (the next thermostat half step has already run)");
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_set_step_size(tgnh_handle h, double dt) {
    CHECK_H(h);
    if (!(dt > 0) || !std::isfinite(dt)) return fail(TGNH_ERR_ARG, "step size must be positive");
    if (dt != h->d.step_size) { tgnh_status rc = deferred_guard(h, "tgnh_set_step_size"); if (rc) return rc; }
    h->d.step_size = dt;
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_set_drude_steps_per_real_step(tgnh_handle h, int n) {
    CHECK_H(h);
    if (n < 1) return fail(TGNH_ERR_ARG, "drudeStepsPerRealStep must be >= 1");
    if (n != h->d.drude_steps_per_real_step) { tgnh_status rc = deferred_guard(h, "tgnh_set_drude_steps_per_real_step"); if (rc) return rc; }
    h->d.drude_steps_per_real_step = n;
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_set_max_drude_distance(tgnh_handle h, double dist) {
    CHECK_H(h);
    if (!(dist >= 0)) return fail(TGNH_ERR_ARG, "setMaxDrudeDistance: Distance cannot be negative");   // API :98-99 (NaN neither)
    h->d.max_drude_distance = dist;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_get_local_dof_terms(tgnh_handle h, double* terms, int* count) {
    CHECK_H(h);
    if (count) *count = h->L.NT;
    if (terms) std::copy(h->local_terms.begin(), h->local_terms.end(), terms);
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_set_global_dof_terms(tgnh_handle h, const double* terms, int count) {
    CHECK_H(h);
    if (!terms || count != h->L.NT) return fail(TGNH_ERR_ARG, "dof term count mismatch");
    if (h->step_count != 0) return fail(TGNH_ERR_STATE, "global dof must be set before the first step");
    if (!h->host_only) HIP_OK(hipSetDevice(h->device));
    h->global_terms.assign(terms, terms + count);
    return finalize_thermostat(h);
}
static tgnh_status materialize_chain(tgnh_handle h, hipStream_t s);

// ---------------------------------------------------------------------------
// mailbox exchange (SURVEY 8e done with stores over xGMI; protocol in tgnh_internal.h)
// ---------------------------------------------------------------------------
extern "C" tgnh_status tgnh_exchange_create(tgnh_handle h, int world, int rank, void* ipc_handle_out, void** mailbox_out) {
    CHECK_H(h);
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle");
    if (world < 1 || world > XCHG_MAX_WORLD || rank < 0 || rank >= world) return fail(TGNH_ERR_ARG, "bad world / rank");
    if (h->L.NT > XCHG_NT_PAD || h->gather_chain) return fail(TGNH_ERR_UNSUPPORTED, "too many thermostats for a mailbox (more than 32 temperature groups: use an all-reduce hook or RCCL)");
    if (h->x_mailbox) return fail(TGNH_ERR_STATE, "exchange already created");
    HIP_OK(hipSetDevice(h->device));
    const size_t bytes = XCHG_MAILBOX_BYTES(world);
    void* p = nullptr;
    HIP_OK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
    h->x_mailbox = static_cast<unsigned long long*>(p);
    HIP_OK(hipMemset(h->x_mailbox, 0, bytes));
    HIP_OK(hipMalloc(&h->d_x_seq, sizeof(unsigned long long)));
    HIP_OK(hipMemset(h->d_x_seq, 0, sizeof(unsigned long long)));
    HIP_OK(hipMalloc(&h->d_x_dead, sizeof(unsigned int)));
    HIP_OK(hipMemset(h->d_x_dead, 0, sizeof(unsigned int)));
    HIP_OK(hipMalloc(&h->d_x_peers, sizeof(unsigned long long*) * world));
    HIP_OK(hipMalloc(&h->d_x_stat, 3 * sizeof(unsigned long long)));
    HIP_OK(hipMemset(h->d_x_stat, 0, 3 * sizeof(unsigned long long)));
    HIP_OK(hipDeviceSynchronize());
    h->x_world = world; h->x_rank = rank;
    if (ipc_handle_out) {
        static_assert(sizeof(hipIpcMemHandle_t) == TGNH_XCHG_HANDLE_BYTES, "IPC handle size");
        hipIpcMemHandle_t ih;
        HIP_OK(hipIpcGetMemHandle(&ih, h->x_mailbox));
        std::memcpy(ipc_handle_out, &ih, sizeof(ih));
    }
    if (mailbox_out) *mailbox_out = h->x_mailbox;
    return TGNH_OK;
}

static tgnh_status exchange_finish_attach(tgnh_handle h, const std::vector<unsigned long long*>& peers) {
    HIP_OK(hipMemcpy(h->d_x_peers, peers.data(), sizeof(unsigned long long*) * peers.size(), hipMemcpyHostToDevice));
    h->x = XchgArgs{};
    h->x.on = 1; h->x.world = h->x_world; h->x.rank = h->x_rank;
    h->x.peers = h->d_x_peers; h->x.mine = h->x_mailbox; h->x.seq = h->d_x_seq; h->x.dead = h->d_x_dead;
    h->x.status = h->d_status;
    h->x.stat = h->d_x_stat;
    h->xchg_on = true;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_exchange_attach(tgnh_handle h, const void* ipc_handles) {
    CHECK_H(h);
    if (!h->x_mailbox) return fail(TGNH_ERR_STATE, "tgnh_exchange_create first");
    if (!ipc_handles) return fail(TGNH_ERR_ARG, "null handles");
    tgnh_status rc = deferred_guard(h, "tgnh_exchange_attach"); if (rc) return rc;
    h->ke_carry = false;
    HIP_OK(hipSetDevice(h->device));
    std::vector<unsigned long long*> peers(h->x_world, nullptr);
    for (int r = 0; r < h->x_world; r++) {
        if (r == h->x_rank) { peers[r] = h->x_mailbox; continue; }
        hipIpcMemHandle_t ih;
        std::memcpy(&ih, static_cast<const char*>(ipc_handles) + (size_t)r * TGNH_XCHG_HANDLE_BYTES, sizeof(ih));
        void* p = nullptr;
        HIP_OK(hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess));
        h->x_opened.push_back(p);
        peers[r] = static_cast<unsigned long long*>(p);
    }
    return exchange_finish_attach(h, peers);
}

extern "C" tgnh_status tgnh_exchange_attach_pointers(tgnh_handle h, void* const* mailboxes) {
    CHECK_H(h);
    if (!h->x_mailbox) return fail(TGNH_ERR_STATE, "tgnh_exchange_create first");
    if (!mailboxes) return fail(TGNH_ERR_ARG, "null mailboxes");
    tgnh_status rc = deferred_guard(h, "tgnh_exchange_attach_pointers"); if (rc) return rc;
    h->ke_carry = false;
    HIP_OK(hipSetDevice(h->device));
    std::vector<unsigned long long*> peers(h->x_world, nullptr);
    for (int r = 0; r < h->x_world; r++) {
        peers[r] = r == h->x_rank ? h->x_mailbox : static_cast<unsigned long long*>(mailboxes[r]);
        if (!peers[r]) return fail(TGNH_ERR_ARG, "null mailbox pointer");
    }
    return exchange_finish_attach(h, peers);
}

extern "C" tgnh_status tgnh_exchange_detach(tgnh_handle h) {
    CHECK_H(h);
    if (!h->xchg_on) return TGNH_OK;
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipDeviceSynchronize());
    {   // a time-out that nobody has asked about yet
        uint32_t f = 0;
        HIP_OK(hipMemcpy(&f, h->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost));
        note_status(h, f);
    }
    if (h->end_pending && !h->failed_code) {          // settle collectively: every rank detaches at the same step
        tgnh_status rc = settle_end(h, (hipStream_t)0); if (rc) return rc;
        HIP_OK(hipDeviceSynchronize());
    }
    if (h->chain_pending && h->xwait_pending && !h->failed_code) {       // an exchange is half done (sent, not yet waited for): finish it
        tgnh_status rc = materialize_chain(h, (hipStream_t)0); if (rc) return rc;
        HIP_OK(hipDeviceSynchronize());
    }
    h->xchg_on = false;
    for (void* p : h->x_opened) (void)hipIpcCloseMemHandle(p);      // the peers' mailboxes; mine stays until tgnh_destroy
    h->x_opened.clear();
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_set_allreduce(tgnh_handle h, tgnh_allreduce_fn fn, void* user) {
    CHECK_H(h);
    tgnh_status rc = deferred_guard(h, "tgnh_set_allreduce"); if (rc) return rc;
    h->ke_carry = false;
    if (h->rccl_comm) { rc = tgnh_rccl_shutdown(h); if (rc) return rc; }
    h->allreduce = fn; h->allreduce_user = user;
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// RCCL: the all-reduce of SURVEY 8e enqueued by the library itself
// ---------------------------------------------------------------------------
// RCCL is bound at the first tgnh_rccl_* call, not at load time: the process may already hold one (PyTorch ships its own
// librccl.so.1 and a second copy in one address space is two sets of communicators), so the copy that is loaded is used, and
// the system's (/opt/rocm/lib) is opened only when there is none.  A library that never shards never touches RCCL.
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool ok = false;
};
void rccl_bind(Rccl& r);
Rccl& rccl() {                                   // (handles of different threads may reach this at once: bound exactly once)
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_bind(r); });
    return r;
}
void rccl_bind(Rccl& r) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the copy the process already has
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { r.error = std::string("RCCL not found (librccl.so.1): ") + dlerror(); return; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
    if (!r.ok) r.error = "RCCL: a required symbol is missing from librccl";
}
#define RCCL_OK(expr)                                                                                        \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return fail(TGNH_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

// the tgnh_allreduce_fn the library installs for itself: the kinetic-energy sums, in place, on the step's stream
int rccl_allreduce(void* buf, int count, void* stream, void* user) {
    tgnh_context* h = static_cast<tgnh_context*>(user);
    return rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, static_cast<ncclComm_t>(h->rccl_comm),
                            static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}
}  // namespace

extern "C" tgnh_status tgnh_rccl_unique_id(void* id_out) {
    if (!id_out) return fail(TGNH_ERR_ARG, "null id");
    static_assert(sizeof(ncclUniqueId) == TGNH_RCCL_ID_BYTES, "ncclUniqueId size");
    if (!rccl().ok) return fail(TGNH_ERR_HIP, rccl().error);
    ncclUniqueId id;
    RCCL_OK(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_set_rccl_comm(tgnh_handle h, void* nccl_comm) {
    CHECK_H(h);
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle");
    tgnh_status rc = deferred_guard(h, "tgnh_set_rccl_comm"); if (rc) return rc;
    h->ke_carry = false;
    if (!rccl().ok) return fail(TGNH_ERR_HIP, rccl().error);
    if (h->rccl_comm) { rc = tgnh_rccl_shutdown(h); if (rc) return rc; }
    if (!nccl_comm) {                                   // (a hook the caller installed with tgnh_set_allreduce is not this call's to clear)
        if (h->allreduce == rccl_allreduce) { h->allreduce = nullptr; h->allreduce_user = nullptr; }
        return TGNH_OK;
    }
    h->rccl_comm = nccl_comm; h->rccl_owned = false;
    h->allreduce = rccl_allreduce; h->allreduce_user = h;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_rccl_init(tgnh_handle h, int world, int rank, const void* id) {
    CHECK_H(h);
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle");
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(TGNH_ERR_ARG, "bad world / rank / id");
    tgnh_status rc = deferred_guard(h, "tgnh_rccl_init"); if (rc) return rc;
    h->ke_carry = false;
    if (!rccl().ok) return fail(TGNH_ERR_HIP, rccl().error);
    if (h->rccl_comm) { rc = tgnh_rccl_shutdown(h); if (rc) return rc; }
    HIP_OK(hipSetDevice(h->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    RCCL_OK(rccl().CommInitRank(&comm, world, uid, rank));
    h->rccl_comm = comm; h->rccl_owned = true;
    h->allreduce = rccl_allreduce; h->allreduce_user = h;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_rccl_shutdown(tgnh_handle h) {
    CHECK_H(h);
    if (!h->rccl_comm) return TGNH_OK;
    if (!h->host_only) { HIP_OK(hipSetDevice(h->device)); HIP_OK(hipDeviceSynchronize()); }
    if (h->allreduce == rccl_allreduce) { h->allreduce = nullptr; h->allreduce_user = nullptr; }
    ncclComm_t comm = static_cast<ncclComm_t>(h->rccl_comm);
    const bool owned = h->rccl_owned;
    h->rccl_comm = nullptr; h->rccl_owned = false;
    if (owned) RCCL_OK(rccl().CommDestroy(comm));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_get_resident_work_groups(tgnh_handle h, int* per_compute_unit) {
    CHECK_H(h);
    if (!per_compute_unit) return fail(TGNH_ERR_ARG, "null out");
    *per_compute_unit = resident_now(h) ? std::max(h->resident_per_cu, h->wresident_per_cu) : 0;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_exchange_wait_stats(tgnh_handle h, void* stream, double* mean_us, double* max_us, int64_t* exchanges) {
    CHECK_H(h);
    if (!h->d_x_stat) return fail(TGNH_ERR_STATE, "tgnh_exchange_create first");
    HIP_OK(hipSetDevice(h->device));
    unsigned long long st[3] = {0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    HIP_OK(hipMemcpyAsync(st, h->d_x_stat, sizeof(st), hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemsetAsync(h->d_x_stat, 0, sizeof(st), s));
    HIP_OK(hipStreamSynchronize(s));
    const double tick_us = 0.01;                           // wall_clock64: 100 MHz
    if (mean_us) *mean_us = st[2] ? tick_us * (double)st[0] / (double)st[2] : 0.0;
    if (max_us) *max_us = tick_us * (double)st[1];
    if (exchanges) *exchanges = (int64_t)st[2];
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_set_resident_share(tgnh_handle h, int share) {
    CHECK_H(h);
    if (share < 1 || share > 64) return fail(TGNH_ERR_ARG, "resident share must be 1..64");
    h->resident_share = share;
    h->wresident_grid = 0;
    for (auto& g : h->resident_grid) g[0] = g[1] = 0;
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// device-reported failures
// ---------------------------------------------------------------------------
constexpr int64_t STATUS_POLL_EVERY = 64;

// Called wherever the status word has reached the host.  bit 2: a mailbox exchange timed out -- from then on the
// kinetic-energy sums are incomplete and the ranks' thermostats diverge; bit 0 in dualNH mode: the Reference platform
// throws (Ref :311-312).  Both make every later step / query fail (entry()).  bit 1 (the harness SHAKE did not
// converge) is reported by tgnh_get_status_flags only: OpenMM's own constraint kernels do not throw either.
static void note_status(tgnh_handle h, uint32_t flags) {
    if (h->failed_code) return;
    // bit 3 first: when step_kernel's work-group 0 gives up on a row it sets bit 3 and withholds the sums, and every other
    // work-group then runs into its own time-out (bit 2) -- a residency problem, not a link fault
    if (flags & 8u) {
        h->failed_code = TGNH_ERR_STATE;
        h->failed = "resident step (noticed at step " + std::to_string((long long)h->step_count) + "): the launch's work-groups did "
                    "not all become resident within the time limit (TGNH_FLAG_RESIDENT_STEP needs the device to itself)";
        if (flags & 4u) h->failed += "; the waiting work-groups timed out in turn (status bits 2 and 3)";
    } else if (flags & 16u) {
        h->failed_code = TGNH_ERR_STATE;
        h->failed = "kinetic-energy pass (noticed at step " + std::to_string((long long)h->step_count) + "): work-group 0 did not "
                    "receive every work-group's row of sums within the time limit (the sums were left as NaN: nothing integrated "
                    "on with a partial sum); the device is shared with something that keeps this launch's work-groups from running "
                    "-- or a chain was handed a NaN kinetic energy: with an all-reduce attached, the rank on which that happened";
    } else if (flags & 4u) {
        h->failed_code = TGNH_ERR_STATE;
        h->failed = "mailbox exchange timed out (noticed at step " + std::to_string((long long)h->step_count) +
                    "): a peer did not send its kinetic-energy sums; the run cannot continue";
    } else if ((flags & 1u) && h->d.mode == TGNH_MODE_DUALNH) {
        h->failed_code = TGNH_ERR_HARDWALL;
        h->failed = "Drude particle moved too far beyond hard wall constraint";        // Ref :311-312
    }
}

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
struct Timed {
    tgnh_context* c; hipStream_t s; int kid; tgnh_context::Ev* ev = nullptr;
    Timed(tgnh_context* c_, hipStream_t s_, int kid_) : c(c_), s(s_), kid(kid_) {
        if (!c->timing) return;
        if (c->timing_only >= 0 && kid != c->timing_only) return;
        if (c->ev_used == c->ev_pool.size()) {
            tgnh_context::Ev e; e.kid = kid;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
            c->ev_pool.push_back(e);
        }
        ev = &c->ev_pool[c->ev_used++];
        ev->kid = kid;
        (void)hipEventRecord(ev->a, s);
    }
    ~Timed() { if (ev) (void)hipEventRecord(ev->b, s); }
};

static tgnh_status need_buffers(tgnh_handle h) {
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle (device -1): no GPU work can be launched on it");
    if (!h->velm) return fail(TGNH_ERR_STATE, "tgnh_bind_buffers has not been called");
    return TGNH_OK;
}

static TileArgs tile_args(tgnh_handle h, const double* scale) {
    TileArgs a{};
    a.posq = h->posq; a.posq_corr = h->posq_corr; a.velm = h->velm;
    a.force = reinterpret_cast<const long long*>(h->force); a.pos_delta = h->pos_delta;
    a.meta = h->d_meta; a.tile_start = h->d_tile_start; a.tile_res = h->d_tile_res; a.res_table = h->d_res_table;
    a.big_com = h->d_big_com;
    a.scale = scale ? scale : h->d_state + h->L.off_scale;
    a.partials = h->d_partials; a.status = h->d_status;
    a.num_tiles = h->num_tiles; a.padded = h->d.padded_num_particles; a.num_groups = h->L.G;
    a.reverse = h->sweep_reverse;
    a.wave_tile = h->d_wave_tile; a.wmeta = h->d_wmeta; a.num_wtiles = h->num_wtiles;
    a.tile_pat = h->d_tile_pat; a.pattern = h->d_pattern; a.wpattern = h->d_wpattern;
    a.use_com = (h->d.mode == TGNH_MODE_TGNH && h->d.use_com_temp_group) ? 1 : 0;
    a.hardwall = h->d.max_drude_distance > 0 ? 1 : 0;                         // Ref :299, Cu :372
    a.dt = h->d.step_size; a.max_dist = h->d.max_drude_distance;
    a.hw_scale = std::sqrt(h->d.kB * h->d.drude_temperature);                 // Ref :300, Cu :299
    return a;
}

// persistent grid = the work-groups of this instantiation that are resident at once (occupancy x CUs), so every
// work-group walks the same number of tiles (+-1) and there is no partial last wave of work-groups
static int grid_for(tgnh_handle h, int ops, bool hardwall, size_t lds) {
    if (h->grid_override > 0) return std::min(h->num_tiles, h->grid_override);
    const int key = ops | (hardwall ? 1 << 16 : 0);
    auto it = h->grid_cache.find(key);
    if (it != h->grid_cache.end()) return it->second;
    int per_cu = tile_blocks_per_cu(h->d.precision, ops, h->gb, lds, (ops & OP_SCALE) && h->inline_chain && h->L.C > 1);
    if (per_cu < 1) per_cu = 2;
    int g = std::min(std::min(h->num_tiles, per_cu * h->num_cus), GRID_CAP);
    if (g < 1) g = 1;
    h->grid_cache[key] = g;
    return g;
}

// wke_kernel: the resident work-groups, at most one per four wavefront tiles
static int wave_grid_for(tgnh_handle h, int ops) {
    const int nw = h->num_wtiles, need = (nw + TBLOCK / 64 - 1) / (TBLOCK / 64);
    if (h->grid_override > 0) return std::max(1, std::min(need, h->grid_override));
    const int key = ops | (1 << 17);
    auto it = h->grid_cache.find(key);
    if (it != h->grid_cache.end()) return it->second;
    int per_cu = wke_blocks_per_cu(h->d.precision, ops, h->gb);
    if (per_cu < 1) per_cu = 2;
#ifdef TGNH_TUNING
    if (const char* e = getenv("TGNH_WKE_PER_CU")) { int v = atoi(e); if (v >= 1) per_cu = std::min(per_cu, v); }
#endif
    int g = std::max(1, std::min(std::min(need, per_cu * h->num_cus), GRID_CAP));
    h->grid_cache[key] = g;
    return g;
}

static ChainArgs chain_args(tgnh_handle h);
static tgnh_status commit_stage(tgnh_handle h, hipStream_t s);

// words of the tagged-row area (TileArgs::rows) and what a launch of `grid` work-groups with NT thermostats writes there
// (row_word in tgnh_kernels.hip: rows in blocks of 64, word-major inside a block, two words per thermostat)
static size_t tagged_words_allocated() { return (size_t)2 * GRID_CAP * CHAIN_INLINE_SUM_NT; }
static size_t tagged_words_touched(int grid, int NT) {
    if (grid < 1) return 0;
    const int r = grid - 1, j = 2 * NT - 1;
    return ((size_t)(r >> 6) * (2 * CHAIN_INLINE_SUM_NT) + j) * 64 + (r & 63) + 1;
}

// The sizes a streaming launch is bound by, checked on the host before it goes out: one row of partial sums per work-group
// in a table of GRID_CAP rows; tagged rows only with G <= 8; a wave-tile table of num_wtiles + 1 entries in which every tile
// holds <= 64 slots (tgnh_create built them so: this is the launch-side half of that contract).
static tgnh_status check_launch(tgnh_handle h, const TileArgs& a, int grid, int block, bool wave, bool tagged) {
    if (grid < 1 || grid > GRID_CAP || grid > h->grid) return fail(TGNH_ERR_STATE, "internal: grid exceeds the partial-row table");
    if (wave) {
        if (!a.wave_tile || (int)h->wave_tile.size() != a.num_wtiles + 1 || a.num_wtiles < 1)
            return fail(TGNH_ERR_STATE, "internal: wave-tile table does not match the launch");
        if ((long long)grid * (block / 64) > (long long)a.num_wtiles + (block / 64) - 1)
            return fail(TGNH_ERR_STATE, "internal: more work-groups than wave tiles");
    } else if (grid > h->num_tiles) return fail(TGNH_ERR_STATE, "internal: more work-groups than tiles");
    if (tagged) {
        if (!a.rows || !a.sync || h->L.NT > CHAIN_INLINE_SUM_NT || tagged_words_touched(grid, h->L.NT) > tagged_words_allocated())
            return fail(TGNH_ERR_STATE, "internal: tagged rows do not fit their area");
    }
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_get_launch_bounds(tgnh_handle h, int32_t out[8]) {
    CHECK_H(h);
    if (!out) return fail(TGNH_ERR_ARG, "null out");
    const int nw = h->num_wtiles;
    int g = h->num_tiles;                                                    // tile_kernel / step_kernel: at most one work-group per tile
    if (nw > 0) g = std::max(g, (nw + TBLOCK / 64 - 1) / (TBLOCK / 64));     // wke_kernel: per four wave tiles (wstep_kernel: per eight)
    g = std::max(1, std::min(g, GRID_CAP));
    const bool tagged = h->L.NT <= CHAIN_INLINE_SUM_NT && (h->d_rows != nullptr || h->host_only);
    out[0] = h->num_tiles; out[1] = nw; out[2] = (int)h->wave_tile.size(); out[3] = g;
    out[4] = h->grid; out[5] = tagged ? (int)tagged_words_allocated() : 0;
    out[6] = tagged ? (int)tagged_words_touched(g, h->L.NT) : 0; out[7] = h->L.NT;
    return TGNH_OK;
}

static tgnh_status run_big_com(tgnh_handle h, bool kick, hipStream_t s) {
    BigComArgs b{};
    b.table = h->d_big_table; b.n = h->num_big; b.velm = h->velm;
    b.force = reinterpret_cast<const long long*>(h->force); b.padded = h->d.padded_num_particles;
    b.kick = kick ? 1 : 0; b.dt = h->d.step_size;
    b.big_com = h->d_big_com;
    b.partials = h->d_partials + (size_t)GRID_CAP * h->L.NT; b.NT = h->L.NT; b.G = h->L.G;
    Timed t(h, s, KID_OTHER);
    HIP_OK(launch_big_com(h->d.precision, b, s));
    return TGNH_OK;
}

// ---- the gather path (tgnh_gather.hip): the same operation masks, by global index ----
static GatherArgs gather_args(tgnh_handle h, const double* scale) {
    GatherArgs a{};
    a.posq = h->posq; a.posq_corr = h->posq_corr; a.velm = h->velm;
    a.force = reinterpret_cast<const long long*>(h->force); a.pos_delta = h->pos_delta;
    a.group = h->d_g_group; a.resid = h->d_g_resid;
    a.res_table = h->d_g_res_table; a.partner = h->d_g_partner; a.com = h->d_g_com;
    a.scale = scale ? scale : h->d_state + h->L.off_scale;
    a.partials = h->d_partials; a.status = h->d_status;
    a.n = h->d.num_particles; a.padded = h->d.padded_num_particles;
    a.com_lanes = h->g_com_lanes;
    a.use_com = (h->d.mode == TGNH_MODE_TGNH && h->d.use_com_temp_group) ? 1 : 0;
    a.n_res = a.use_com ? (int)h->g_res_table.size() : 0;
    a.G = h->L.G; a.NT = h->L.NT;
    a.hardwall = h->d.max_drude_distance > 0 ? 1 : 0;                         // Ref :299, Cu :372
    a.dt = h->d.step_size; a.max_dist = h->d.max_drude_distance;
    a.hw_scale = std::sqrt(h->d.kB * h->d.drude_temperature);                 // Ref :300, Cu :299
    return a;
}

// One operation mask of run_tile as launches of the gather kernels: the velocity / position part (rescale, kick, drift,
// posDelta, move, hard wall) first, the kinetic energies of what it stored after it (K's order: Cu :384-388 then :474-488).
static tgnh_status run_gather(tgnh_handle h, int ops, int kid, hipStream_t s, const double* scale) {
    GatherArgs a = gather_args(h, scale);
    if ((ops & (OP_POSDELTA | OP_MOVE)) && !h->pos_delta) return fail(TGNH_ERR_STATE, "posDelta buffer not bound");
    const int upd = ops & (OP_SCALE | OP_KICK | OP_DRIFT | OP_POSDELTA | OP_MOVE | OP_PREKICK);
    if (ops & OP_NOSTORE) return fail(TGNH_ERR_STATE, "internal: the gather path stores every kick");
    Timed t(h, s, kid);
    if (upd) {
        // K :474-479 before :351-353: v - v_com of the velocities about to be rescaled.  On this path every rescale follows a
        // kinetic-energy pass and its chain inside one entry point (the flags that would part them are ignored), so the table
        // that pass left is of these very velocities: not computed again
        if ((upd & OP_SCALE) && a.use_com && !(h->g_com_fresh && !(upd & OP_PREKICK))) {
            a.kick_com = (upd & OP_PREKICK) ? 1 : 0;
            HIP_OK(launch_gather_com(h->d.precision, a, s));
        }
        h->g_com_fresh = false;
        a.ops = upd;
        HIP_OK(launch_gather_update(h->d.precision, a, s));
    }
    if (ops & OP_KE) {
        a.kick_com = 0;
        if (a.use_com) HIP_OK(launch_gather_com(h->d.precision, a, s));
        const int grid = gather_ke_grid(a);
        HIP_OK(launch_gather_ke(h->d.precision, a, grid, s));
        h->ke_parts = grid;
        h->tail_summed = false;
        h->g_com_fresh = a.use_com != 0;
    }
    return TGNH_OK;
}

// sum the rows of the gather path's kinetic-energy kernel [+ all-reduce], run the chain: more than 34 thermostats / long links
static tgnh_status run_chain_gather(tgnh_handle h, hipStream_t s, bool sum_only) {
    ChainArgs a = chain_args(h);
    Timed t(h, s, KID_CHAIN);
    HIP_OK(launch_gather_rowsum(h->d_partials, h->ke_parts, h->L.NT, h->d_state + h->L.off_ke_red, s));
    if (h->allreduce && h->allreduce(h->d_state + h->L.off_ke_red, h->L.NT, (void*)s, h->allreduce_user) != 0)
        return fail(TGNH_ERR_HIP, "all-reduce hook failed");
    if (!sum_only) HIP_OK(launch_gather_chain(a, h->d_g_scratch, s));
    return TGNH_OK;
}

static tgnh_status run_tile(tgnh_handle h, int ops, int kid, hipStream_t s, const double* scale = nullptr) {
    if (h->generic) return run_gather(h, ops, kid, s, scale);
    TileArgs a = tile_args(h, scale);
    bool inline_chain = false;
    bool pingpong = false;
    if ((ops & OP_SCALE) && h->chain_pending && !scale) {           // this rescale launch runs the chain itself
        // A carried chain (no KE pass before it: nothing has committed the staged block on the way) reads the thermostat where the
        // last in-kernel chain left it and writes the other copy: the two blocks differ only in what a chain writes, and a
        // chain writes all of that every time (eta, etaDot, etaDotDot, KE before / after, the scale factors, KESum)
        pingpong = h->carry_pending && h->stage_pending;
        if (h->stage_pending && !pingpong) { tgnh_status rc = commit_stage(h, s); if (rc) return rc; }
        a.chain_on = 1;
        a.chain = chain_args(h);                                    // (takes note of the staged block: cleared there)
        a.chain.chain_twice = h->chain_pending_twice ? 1 : 0;
        a.chain.ke_carry = h->carry_pending ? 1 : 0;
        a.sum_rows = h->sum_pending ? (h->ke_parts + h->num_big <= h->inline_sum_rows ? 1 : 2) : 0;
        a.x_wait = h->xwait_pending ? 1 : 0;
        a.st_in = pingpong ? h->d_stage : h->d_state;
        a.st_out = pingpong ? h->d_state : h->d_stage;
        inline_chain = true;
    }
    if ((ops & (OP_POSDELTA | OP_MOVE)) && !h->pos_delta) return fail(TGNH_ERR_STATE, "posDelta buffer not bound");
    size_t lds = tile_lds_bytes(h->d.precision, ops, a.hardwall != 0, a.use_com != 0);
    if ((ops & OP_KE) && h->gb == 0) lds += sizeof(double) * (TBLOCK / 64) * h->L.G;   // per-wave group bins
    // the pure KE passes (KE, kick+KE, kick+KE unstored) run over the wave tiles when the topology has them
    const bool wave = h->wave_ke && (ops & OP_KE) && !(ops & ~(OP_KE | OP_KICK | OP_NOSTORE));
    const int grid = wave ? wave_grid_for(h, ops) : grid_for(h, ops, a.hardwall != 0, lds);
    if ((ops & OP_KE) && h->stage_pending && !inline_chain) {      // commit the staged thermostat block on the way
        a.commit_len = h->L.total; a.commit_src = h->d_stage; a.commit_dst = h->d_state;
        a.commit_skip = h->L.off_ke_red; a.commit_skip_n = h->L.NT;
        h->stage_pending = false;
    }
    if (ops & OP_KE) {
        // wave tiles: where a launch that only sums the partial rows would follow (an all-reduce waits for the sums, or the
        // system is too large for the next rescale launch to sum them in its prologue), work-group 0 of this launch does it
        h->tail_summed = wave && h->d_rows && !h->xchg_on && h->L.NT <= CHAIN_INLINE_SUM_NT &&
                         !(h->inline_chain && !h->allreduce && (grid + h->num_big <= h->inline_sum_rows || h->inline_sum_all));
        if (h->tail_summed) { a.tail_sum = 1; a.rows = h->d_rows; a.sync = h->d_sync; a.ke_red = h->d_state + h->L.off_ke_red; }
        h->ke_parts = grid;
        if (h->num_big && a.use_com) {
            // COM velocity of every big molecule for the velocities this launch reduces: the current ones, or the
            // kicked ones (the kick is linear, so sum m v' = sum (m v + dt/2 F) needs no second pass).  The rescale
            // launches that follow reuse the table: velocities do not change between a KE pass and its rescale.
            tgnh_status rc = run_big_com(h, (ops & OP_KICK) != 0, s); if (rc) return rc;
        }
    }
    { tgnh_status rc = check_launch(h, a, grid, TBLOCK, wave, a.tail_sum != 0); if (rc) return rc; }
    {
        Timed t(h, s, kid);
        if (wave) HIP_OK(launch_wke(h->d.precision, ops, h->gb, a, grid, s));
        else HIP_OK(launch_tile(h->d.precision, ops, h->gb, a, grid, lds, s));
    }
    if (inline_chain) {            // the advanced thermostat now lies in d_stage (carried from a staged block: back in d_state)
        h->chain_pending = false; h->sum_pending = false; h->xwait_pending = false; h->carry_pending = false; h->stage_pending = !pingpong;
    }
    if (h->alternate_sweeps) h->sweep_reverse ^= 1;      // the next streaming launch starts where this one ends
    return TGNH_OK;
}

static ChainArgs chain_args(tgnh_handle h) {
    ChainArgs a{};
    a.L = h->L; a.st = h->d_state; a.partials = h->d_partials; a.nparts = h->ke_parts;
    a.nbig = h->num_big;
    a.dt = h->d.step_size; a.S = h->d.drude_steps_per_real_step;
    a.dtc = a.dt / a.S; a.inv_dtc = 1.0 / a.dtc;                               // Cu :440-443
    a.realkbT = h->realkbT; a.drudekbT = h->drudekbT;
    // chains of 5-16 links: chain_long_kernel<C>, the links in registers (round 3 ran them a link per lane, chain_lanes_run: kept
    // behind a switch for the comparison in profiles/r04_chain_cost.md)
    a.lanes = 0;
#ifdef TGNH_TUNING
    if (const char* e = getenv("TGNH_CHAIN_LANES")) a.lanes = e[0] != '0';
#endif
    a.stage = h->d_stage;
    a.status = h->d_status;
    if (h->xchg_on) a.x = h->x;
    a.commit = h->stage_pending ? 1 : 0;     // every chain_kernel launch takes over a staged block first
    h->stage_pending = false;
    return a;
}

// a staged block with no chain_kernel launch coming up: commit it by a launch that does nothing else
static tgnh_status commit_stage(tgnh_handle h, hipStream_t s) {
    if (!h->stage_pending) return TGNH_OK;
    ChainArgs a = chain_args(h);
    a.do_sum = 0; a.do_chain = 0;
    HIP_OK(launch_chain(a, s));
    return TGNH_OK;
}

// sum the work-group partials, all-reduce across ranks when sharded, run the chain
static tgnh_status run_chain(tgnh_handle h, hipStream_t s, bool twice) {
    if (h->gather_chain) return run_chain_gather(h, s, false);
    ChainArgs a = chain_args(h);
    a.chain_twice = twice ? 1 : 0;
    if (h->xchg_on) {                // sharded, mailbox exchange: the sum launch sends; whoever runs the chain waits
        a.do_sum = 1; a.x_send = 1;
        if (h->inline_chain) {
            a.do_chain = 0;
            { Timed t(h, s, KID_CHAIN); HIP_OK(launch_chain(a, s)); }
            h->chain_pending = true; h->chain_pending_twice = twice; h->xwait_pending = true;
        } else {
            a.do_chain = 1; a.x_wait = 1;
            Timed t(h, s, KID_CHAIN);
            HIP_OK(launch_chain(a, s));
        }
        return TGNH_OK;
    }
    if (h->inline_chain && !h->allreduce && h->L.NT <= CHAIN_INLINE_SUM_NT &&
        (h->ke_parts + h->num_big <= h->inline_sum_rows || h->inline_sum_all)) {
        // Unsharded, one-link chains, G <= 8: nothing to launch -- the next rescale launch sums the partial rows and
        // runs the chain in its prologue (3 launches per step).  Up to 256 rows its chain wavefront reads them alone
        // (one batch of loads); more rows are read by all four wavefronts, a quarter each, ahead of their tile
        // loads (read by one wavefront they were a chain of L2 misses on the critical path, +7-9 us) -- that up to 2 M
        // slots (inline_sum_all).  +4 % steps/s at 625 k slots, +7-17 % for small systems
        // (profiles/r01_tuning_sweep.log).
        h->chain_pending = true; h->sum_pending = true; h->chain_pending_twice = twice;
        return TGNH_OK;
    }
    const bool summed = h->tail_summed;       // wke_kernel's tail sum: ke_red is complete, no row-sum launch (and nothing staged: a KE launch commits)
    h->tail_summed = false;
    if (h->inline_chain) {           // sum (and all-reduce) now, the chain itself inside the next rescale launch
        a.do_sum = 1; a.do_chain = 0;
        if (!summed) { Timed t(h, s, KID_CHAIN); HIP_OK(launch_chain(a, s)); }
        if (h->allreduce && h->allreduce(h->d_state + h->L.off_ke_red, h->L.NT, (void*)s, h->allreduce_user) != 0)
            return fail(TGNH_ERR_HIP, "all-reduce hook failed");
        h->chain_pending = true; h->chain_pending_twice = twice;
        return TGNH_OK;
    }
    if (h->allreduce) {
        a.do_sum = 1; a.do_chain = 0;
        if (!summed) { Timed t(h, s, KID_CHAIN); HIP_OK(launch_chain(a, s)); }
        if (h->allreduce(h->d_state + h->L.off_ke_red, h->L.NT, (void*)s, h->allreduce_user) != 0)
            return fail(TGNH_ERR_HIP, "all-reduce hook failed");
        a.do_sum = 0; a.do_chain = 1; a.commit = 0;
        { Timed t(h, s, KID_CHAIN); HIP_OK(launch_chain(a, s)); }
    } else {
        a.do_sum = summed ? 0 : 1; a.do_chain = 1;
        Timed t(h, s, KID_CHAIN);
        HIP_OK(launch_chain(a, s));
    }
    return TGNH_OK;
}

// a chain that is still waiting for its rescale launch is run now, in place, by the standalone kernel
static tgnh_status materialize_chain(tgnh_handle h, hipStream_t s) {
    { tgnh_status rc = settle_end(h, s); if (rc) return rc; }
    if (!h->chain_pending) return commit_stage(h, s);
    ChainArgs a = chain_args(h);
    a.do_sum = h->sum_pending ? 1 : 0; a.do_chain = 1; a.chain_twice = h->chain_pending_twice ? 1 : 0;
    a.x_wait = h->xwait_pending ? 1 : 0;
    a.ke_carry = h->carry_pending ? 1 : 0;
    { Timed t(h, s, KID_CHAIN); HIP_OK(launch_chain(a, s)); }
    h->chain_pending = false; h->sum_pending = false; h->xwait_pending = false; h->carry_pending = false;
    return TGNH_OK;
}

// ---- TGNH_FLAG_RESIDENT_STEP: one launch per time step (step_kernel) ----
// Eligible: deferred pass structure, one-link chains (the chain runs inside the launch), at most 8 temperature groups,
// and an exchange the kernel can do itself (none, or the mailboxes -- a collective hook is a launch of its own).
static bool resident_kind(tgnh_handle h, int kind) {
    if (!((h->d.flags & TGNH_FLAG_RESIDENT_STEP) && h->inline_chain && h->gb != 0 && h->L.NT <= CHAIN_INLINE_SUM_NT &&
          (h->xchg_on || !h->allreduce))) return false;
    if (kind == 0 && h->wresident_per_cu > 0) return true;           // wstep_kernel: a whole deferred step, chains of 1-4 links
    return h->resident_per_cu > 0 && h->L.C == 1;                    // step_kernel: every kind, one-link chains
}
static bool resident_now(tgnh_handle h) {
    return resident_kind(h, (h->d.flags & TGNH_FLAG_DEFER_SCALE) ? 0 : 1);
}

extern "C" tgnh_status tgnh_get_step_path(tgnh_handle h, int* gather, const char** reason) {
    CHECK_H(h);
    if (gather) *gather = h->generic ? (h->gather_chain ? 2 : 1) : 0;
    if (reason) *reason = h->generic_reason.c_str();
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_get_resident_kernel(tgnh_handle h, int* which) {
    CHECK_H(h);
    if (!which) return fail(TGNH_ERR_ARG, "null out");
    const int kind = (h->d.flags & TGNH_FLAG_DEFER_SCALE) ? 0 : 1;
    *which = !resident_kind(h, kind) ? 0 : (kind == 0 && h->wresident_per_cu > 0) ? 2 : 1;
    return TGNH_OK;
}

// One launch of step_kernel.  kind 0: a whole deferred step (the last step's end half + this step's begin half, both chain
// halves); 1 / 2: the begin / end half of the reference's pass structure; 3 / 4: the same around the constraint call-outs.
static tgnh_status run_resident(tgnh_handle h, hipStream_t s, int kind) {
    if (h->stage_pending) { tgnh_status rc = commit_stage(h, s); if (rc) return rc; }
    TileArgs a = tile_args(h, nullptr);
    const int ops2 = step_kind_ops2(kind);
    if ((ops2 & OP_POSDELTA) && !h->pos_delta) return fail(TGNH_ERR_STATE, "posDelta buffer not bound");
    const bool hw = a.hardwall != 0 && (ops2 & (OP_DRIFT | OP_MOVE));
    const size_t lds = tile_lds_bytes(h->d.precision, ops2, hw, a.use_com != 0);
    int& grid = h->resident_grid[kind][hw ? 1 : 0];
    if (grid == 0 && !(kind == 0 && h->wresident_per_cu > 0)) {       // the work-groups that are resident at once (counted at create; never more than this kind's own occupancy)
        int per_cu = std::min(h->resident_per_cu, step_blocks_per_cu(h->d.precision, h->gb, kind, lds));
        if (per_cu < 1) return fail(TGNH_ERR_HIP, "step_kernel: occupancy query failed");
        grid = std::max(1, std::min(std::min(h->num_tiles, per_cu * h->num_cus / h->resident_share), GRID_CAP));
    }
    a.chain_on = 1;
    a.chain = chain_args(h);
    a.chain.chain_twice = kind == 0 ? 1 : 0;
    a.chain.nparts = grid;
    a.x_wait = 1;
    if (!h->xchg_on) a.chain.x = h->self_x;            // unsharded: the private one-rank mailbox
    a.st_in = h->d_state; a.st_out = h->d_state;       // advanced in place by work-group 0 after everybody has read it
    a.sync = h->d_sync; a.rows = h->d_rows;
    if (h->num_big && a.use_com) { tgnh_status rc = run_big_com(h, kind == 0 || kind == 2, s); if (rc) return rc; }
    h->last_step_kind = kind;
    if (kind == 0 && h->wresident_per_cu > 0) {            // a whole deferred step over wave tiles (wstep_kernel)
        if (h->wresident_grid == 0) {
            const int need = (h->num_wtiles + WBLOCK / 64 - 1) / (WBLOCK / 64);
            h->wresident_grid = std::max(1, std::min(std::min(need, h->wresident_per_cu * h->num_cus / h->resident_share), GRID_CAP));
        }
        a.chain.nparts = h->wresident_grid;
        h->ke_parts = h->wresident_grid;
        { tgnh_status rc = check_launch(h, a, h->wresident_grid, WBLOCK, true, true); if (rc) return rc; }
        Timed t(h, s, KID_STEP);
        HIP_OK(launch_wstep(h->d.precision, h->gb, h->L.C > 1, a, h->wresident_grid, s));
    } else {
        h->ke_parts = grid;
        { tgnh_status rc = check_launch(h, a, grid, TBLOCK, false, true); if (rc) return rc; }
        Timed t(h, s, KID_STEP);
        HIP_OK(launch_step(h->d.precision, h->gb, kind, a, grid, lds, s));
    }
    // the first pass walked the tiles in direction sweep_reverse, the second one back: the next launch starts here
    h->end_pending = false; h->scale_pending = false; h->kick_pending = false; h->first_half_done = false;
    h->chain_pending = false; h->sum_pending = false; h->xwait_pending = false;
    return TGNH_OK;
}

// make scale[] hold the first thermostat half step for the current velocities (Ref :231, Cu :336)
static tgnh_status first_half(tgnh_handle h, hipStream_t s) {
    if (h->first_half_done) return TGNH_OK;               // DEFER_SCALE: already folded into scale[]
    if (h->ke_carry) {
        // TRUST_STATE_CHANGED, and nothing has written velocities since the last end half's rescale: every kinetic-energy bin
        // of the stored velocities is s^2 times the bin that chain started from -- the ke_post it left (Cu :574 tracks exactly
        // that product) -- so this half's KE pass (Cu :474-488) and its row sum are not run: the chain starts from ke_post
        h->ke_carry = false;
        if (h->num_big && h->d.mode == TGNH_MODE_TGNH && h->d.use_com_temp_group) {
            // (molecules longer than a tile: the KE pass that is not run would have left their centre-of-mass velocities in
            // the table the rescale launch reads -- the end half's rescale has changed them since)
            tgnh_status rc = run_big_com(h, false, s); if (rc) return rc;
        }
        if (h->inline_chain) {                            // ... inside the rescale launch that follows: this half step is ONE launch
            h->chain_pending = true; h->chain_pending_twice = false; h->sum_pending = false; h->carry_pending = true;
            return TGNH_OK;
        }
        ChainArgs a = chain_args(h);
        a.do_sum = 0; a.do_chain = 1; a.ke_carry = 1;
        Timed t(h, s, KID_CHAIN);
        HIP_OK(launch_chain(a, s));
        return TGNH_OK;
    }
    tgnh_status rc = run_tile(h, OP_KE, KID_KE, s); if (rc) return rc;
    return run_chain(h, s, false);
}

// Every entry point that launches work or hands results back starts here: a failure the device reported earlier
// (a mailbox exchange that timed out; in dualNH mode a Drude beyond twice the hard wall, Ref :311-312) is sticky --
// the trajectory is no longer the integrator's, so nothing more is computed on it.
static tgnh_status entry(tgnh_handle h, bool need_bufs) {
    if (!h) return fail(TGNH_ERR_ARG, "null handle");
    h->g_com_fresh = false;                                           // (the caller may have written velocities since the last entry point)
    if (need_bufs) { tgnh_status rc = need_buffers(h); if (rc) return rc; }
    if (!h->host_only) {
        HIP_OK(hipSetDevice(h->device));
        if (h->h_status_seen) note_status(h, *h->h_status_seen);     // the last periodic read-back, if it has landed
    }
    if (h->failed_code) return fail(h->failed_code, h->failed);
    return TGNH_OK;
}

// Every STATUS_POLL_EVERY steps the status word is copied to pinned host memory behind the step (no synchronisation;
// looked at on a later entry): a caller that never asks for anything still learns of a failure within that many steps.
static tgnh_status poll_status_async(tgnh_handle h, hipStream_t s) {
    if (!h->h_status_seen || h->step_count % STATUS_POLL_EVERY != 0) return TGNH_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return TGNH_OK;
    HIP_OK(hipMemcpyAsync(h->h_status_seen, h->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return TGNH_OK;
}

// Consecutive streaming launches sweep the slots in alternating directions (a launch starts where the last one ended), and
// the direction decides the order in which a wavefront adds its tiles' kinetic energies, i.e. the last bits of the sums.  A
// time step holds an odd number of sweeps in every pass structure, so in an undisturbed run step k starts in direction k & 1;
// that is made the rule: whatever was launched between two steps (queries, a flush), a step starts in the direction of
// its number.  The trajectory's bits are then a function of the state and the step counter alone -- a handle restored from a
// checkpoint (tgnh_set_time carries the counter) continues bit for bit, ranks of a sharded run sweep alike.
static void start_of_step(tgnh_handle h) {
    if (h->alternate_sweeps) h->sweep_reverse = (int)(h->step_count & 1);
}

extern "C" tgnh_status tgnh_step_begin(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    start_of_step(h);
    if (h->end_pending) {
        if (resident_now(h)) return run_resident(h, s, 0);           // the last step's end half and this begin half: one launch
        rc = settle_end(h, s); if (rc) return rc;
    }
    if (!(h->d.flags & TGNH_FLAG_DEFER_SCALE) && resident_now(h) && !h->ke_carry)
        return run_resident(h, s, 1);                                // reference pass structure: KE, chain, rescale+kick+drift in one launch
    rc = first_half(h, s); if (rc) return rc;                       // (kinetic energies carried over: no first pass, no meeting -- the tile launch with its in-kernel chain)
    // Cu :351-376 fused; with a half kick still pending from the last step_end (DEFER_SCALE) that kick comes first
    rc = run_tile(h, (h->kick_pending ? OP_PREKICK : 0) | OP_SCALE | OP_KICK | OP_DRIFT, KID_SKD, s); if (rc) return rc;
    h->scale_pending = false; h->kick_pending = false; h->first_half_done = false;
    return TGNH_OK;
}

static tgnh_status second_half(tgnh_handle h, hipStream_t s, int kick_ops) {
    tgnh_status rc;
    const bool defer = (h->d.flags & TGNH_FLAG_DEFER_SCALE) != 0;
    h->ke_carry = false;                                   // (an end half without a begin half before it: its own KE pass runs in any case)
    if (h->scale_pending || h->kick_pending || h->end_pending) { rc = flush_impl(h, s); if (rc) return rc; }   // two end halves in a row
    if (!defer && resident_now(h)) {
        // reference pass structure: kick, KE, chain, rescale (Cu :384-402) in one launch; velocities are final when it ends
        rc = run_resident(h, s, kick_ops ? 2 : 4); if (rc) return rc;
        h->ke_carry = h->carry_ok && !h->allreduce && !h->xchg_on;
        h->time += h->d.step_size;
        h->step_count += 1;
        return poll_status_async(h, s);
    }
    if (kick_ops && resident_now(h)) {
        // TGNH_FLAG_RESIDENT_STEP: nothing is launched here -- the next tgnh_step_begin runs this end half and its own
        // begin half in one launch (step_kernel); anything that needs the state earlier settles it the classic way
        h->end_pending = true;
        h->time += h->d.step_size;
        h->step_count += 1;
        return poll_status_async(h, s);
    }
    // DEFER_SCALE, fused path: the kicked velocities only feed the sums (Cu :384-388 + :474-488); the next step's first
    // launch -- or tgnh_flush -- forms them again from the same force buffer and goes on from there.
    // The reference's own structure, fused path (round 4): the same unstored kick+KE pass, and the rescale launch of THIS call forms
    // the kicked velocities again before it rescales them (OP_PREKICK: the same expression on the same force buffer, the same
    // bits) -- V r, F r | V r/w, F r = 144 B per slot where kick+KE with a store and a plain rescale move 152, and the read-only
    // pass runs at 62 us where the storing one takes 87-92 (5 M slots).  velm holds the reference's end-of-step velocities when
    // tgnh_step_end returns, as before.  (The split path's halves work on stored velocities around the constraint call-outs.)
    const bool fold = !defer && kick_ops != 0 && !h->generic;           // (the gather path stores its kick: K's own structure)
    const int nostore = kick_ops && !h->generic ? OP_NOSTORE : 0;
    h->end_folded = fold;
    rc = run_tile(h, kick_ops | OP_KE | nostore, kick_ops ? KID_KICK_KE : KID_KE, s); if (rc) return rc;
    if (defer) {
        rc = run_chain(h, s, true); if (rc) return rc;
        h->scale_pending = true; h->first_half_done = true; h->kick_pending = nostore != 0;
    } else {
        rc = run_chain(h, s, false); if (rc) return rc;                            // Cu :394-395
        rc = run_tile(h, (fold ? OP_PREKICK : 0) | OP_SCALE, KID_SCALE, s); if (rc) return rc;   // Cu :402 (and :384-388 again, see above)
        // the velocities now stored have the bins ke_post; they stay that until somebody writes velocities (unsharded only: a
        // rank that recomputes while its peers carry over would enter a collective alone)
        h->ke_carry = h->carry_ok && !h->allreduce && !h->xchg_on;
    }
    h->time += h->d.step_size;                                                     // Cu :405-406 ; Ref :413-414
    h->step_count += 1;
    return poll_status_async(h, s);
}

extern "C" tgnh_status tgnh_step_end(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    return second_half(h, (hipStream_t)stream, OP_KICK);
}

// The split entry points work on stored velocities: a deferred half kick is materialised first.
static tgnh_status settle_kick(tgnh_handle h, hipStream_t s) {
    return (h->kick_pending || h->end_pending) ? flush_impl(h, s) : TGNH_OK;
}

// TGNH_FLAG_RESIDENT_STEP left the end half of the last step to the next tgnh_step_begin; somebody needs it now:
// the classic launches (kick+KE without a velocity store, row sum [+ exchange], both chain halves pending)
static tgnh_status settle_end(tgnh_handle h, hipStream_t s) {
    if (!h->end_pending) return TGNH_OK;
    h->end_pending = false;
    tgnh_status rc = run_tile(h, OP_KICK | OP_KE | OP_NOSTORE, KID_KICK_KE, s); if (rc) return rc;
    rc = run_chain(h, s, true); if (rc) return rc;
    h->scale_pending = true; h->first_half_done = true; h->kick_pending = true;
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_step_begin_kick(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = settle_kick(h, s); if (rc) return rc;
    start_of_step(h);
    if (!(h->d.flags & TGNH_FLAG_DEFER_SCALE) && resident_now(h) && !h->ke_carry) return run_resident(h, s, 3);
    rc = first_half(h, s); if (rc) return rc;
    rc = run_tile(h, OP_SCALE | OP_KICK | OP_POSDELTA, KID_OTHER, s); if (rc) return rc;   // Cu :351-360
    h->scale_pending = false; h->first_half_done = false;
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_step_begin_move(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    h->ke_carry = false;
    rc = settle_kick(h, (hipStream_t)stream); if (rc) return rc;
    return run_tile(h, OP_MOVE, KID_OTHER, (hipStream_t)stream);                   // Cu :366-376
}
extern "C" tgnh_status tgnh_step_end_kick(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    h->ke_carry = false;
    rc = settle_kick(h, (hipStream_t)stream); if (rc) return rc;
    return run_tile(h, OP_KICK, KID_OTHER, (hipStream_t)stream);                   // Cu :384-388
}
extern "C" tgnh_status tgnh_step_end_thermo(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    return second_half(h, (hipStream_t)stream, 0);                                 // Cu :394-406
}
extern "C" tgnh_status tgnh_half_kick(tgnh_handle h, void* stream) { return tgnh_step_end_kick(h, stream); }

// velm <- the reference's end-of-step velocities: the pending half kick (same force buffer), then the end-of-step
// factors; scale[] keeps only the pre-run first half of the coming step
static tgnh_status flush_impl(tgnh_handle h, hipStream_t s) {
    tgnh_status rc = settle_end(h, s); if (rc) return rc;
    if (!h->scale_pending && !h->kick_pending) return TGNH_OK;
    rc = materialize_chain(h, s); if (rc) return rc;
    if (h->scale_pending) {
        rc = run_tile(h, (h->kick_pending ? OP_PREKICK : 0) | OP_SCALE, KID_SCALE, s, h->d_state + h->L.off_scale_a); if (rc) return rc;
        HIP_OK(hipMemcpyAsync(h->d_state + h->L.off_scale, h->d_state + h->L.off_scale_b, sizeof(double) * h->L.NT,
                              hipMemcpyDeviceToDevice, s));
    } else {
        rc = run_tile(h, OP_KICK, KID_OTHER, s); if (rc) return rc;
    }
    if (h->num_big && h->d.mode == TGNH_MODE_TGNH && h->d.use_com_temp_group) {
        rc = run_big_com(h, false, s); if (rc) return rc;             // the velocities just changed: refresh the COM table
    }
    h->scale_pending = false; h->kick_pending = false;   // first_half_done stays
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_flush(tgnh_handle h, void* stream) {
    CHECK_H(h);
    if (!h->scale_pending && !h->kick_pending && !h->end_pending) return TGNH_OK;
    tgnh_status rc = entry(h, true); if (rc) return rc;
    return flush_impl(h, (hipStream_t)stream);
}

extern "C" tgnh_status tgnh_note_replayed_steps(tgnh_handle h, int nsteps) {
    CHECK_H(h);
    if (nsteps < 0) return fail(TGNH_ERR_ARG, "negative step count");
    h->time += h->d.step_size * nsteps;
    h->step_count += nsteps;
    return TGNH_OK;
}

// Restores the clock of a checkpointed run (the reference keeps time / stepCount in the platform data, Ref :413-414,
// Cu :405-406, and OpenMM's checkpoints carry them).
extern "C" tgnh_status tgnh_set_time(tgnh_handle h, double time, int64_t step_count) {
    CHECK_H(h);
    if (step_count < 0) return fail(TGNH_ERR_ARG, "negative step count");
    h->time = time;
    h->step_count = step_count;
    return TGNH_OK;
}

static uint32_t owed(tgnh_handle h) {
    return (h->end_pending ? 1u : 0u) | (h->scale_pending ? 2u : 0u) | (h->kick_pending ? 4u : 0u) | (h->first_half_done ? 8u : 0u) |
           (h->chain_pending ? 16u : 0u) | (h->sum_pending ? 32u : 0u) | (h->xwait_pending ? 64u : 0u) | (h->stage_pending ? 128u : 0u);
}
extern "C" tgnh_status tgnh_get_pending_state(tgnh_handle h, uint32_t* bits) {
    CHECK_H(h);
    if (!bits) return fail(TGNH_ERR_ARG, "null out");
    *bits = owed(h) | (h->sweep_reverse ? 256u : 0u) | (h->ke_carry ? 512u : 0u);
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_state_changed(tgnh_handle h) {
    CHECK_H(h);
    tgnh_status rc = deferred_guard(h, "tgnh_state_changed");
    if (rc) return rc;
    h->ke_carry = false;              // TRUST_STATE_CHANGED: the next thermostat half step sums the kinetic energies again (Cu :474-488)
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// queries
// ---------------------------------------------------------------------------
static tgnh_status read_state(tgnh_handle h, int off, int n, hipStream_t s, double* out) {
    if (!out) return fail(TGNH_ERR_ARG, "null out");
    if (h->host_only) { std::copy(h->h_state.begin() + off, h->h_state.begin() + off + n, out); return TGNH_OK; }
    { tgnh_status rc = entry(h, false); if (rc) return rc; }
    { tgnh_status rc = materialize_chain(h, s); if (rc) return rc; }
    HIP_OK(hipMemcpyAsync(out, h->d_state + off, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(h->h_status_seen, h->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    note_status(h, *h->h_status_seen);       // what was just read may come from a failed exchange: say so now
    if (h->failed_code) return fail(h->failed_code, h->failed);
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_get_kinetic_energy(tgnh_handle h, int ke_sum_valid, void* stream, double* out) {
    CHECK_H(h);
    if (!out) return fail(TGNH_ERR_ARG, "null out");
    hipStream_t s = (hipStream_t)stream;
    if (h->d.mode == TGNH_MODE_TGNH && ke_sum_valid)                               // Cu :654-658
        return read_state(h, h->L.off_kesum, 1, s, out);
    tgnh_status rc = entry(h, true); if (rc) return rc;
    rc = flush_impl(h, s); if (rc) return rc;
    const double ts = h->d.mode == TGNH_MODE_DUALNH ? 0.5 * h->d.step_size : 0.0; // Ref :587 ; Cu :656
    HIP_OK(launch_plain_ke(h->d.precision, h->velm, reinterpret_cast<const long long*>(h->force), h->d.num_particles,
                           h->d.padded_num_particles, ts, h->d_scalar, s));
    HIP_OK(hipMemcpyAsync(out, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_get_num_thermostats(tgnh_handle h, int* count) {
    CHECK_H(h);
    if (count) *count = h->L.NT;
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_get_last_kinetic_energies(tgnh_handle h, void* stream, double* ke) {
    CHECK_H(h);
    return read_state(h, h->L.off_ke, h->L.NT, (hipStream_t)stream, ke);
}
extern "C" tgnh_status tgnh_get_last_scale_factors(tgnh_handle h, void* stream, double* scale) {
    CHECK_H(h);
    return read_state(h, h->L.off_scale_a, h->L.NT, (hipStream_t)stream, scale);
}
extern "C" tgnh_status tgnh_get_status_flags(tgnh_handle h, void* stream, uint32_t* flags) {
    CHECK_H(h);
    if (!flags) return fail(TGNH_ERR_ARG, "null flags");
    if (h->host_only) { *flags = 0; return TGNH_OK; }
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipMemcpyAsync(h->h_status_seen, h->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    *flags = *h->h_status_seen;              // always handed back, also beside an error code
    note_status(h, *flags);
    if (h->failed_code) return fail(h->failed_code, h->failed);
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_get_time(tgnh_handle h, double* time, int64_t* step_count) {
    CHECK_H(h);
    if (time) *time = h->time;
    if (step_count) *step_count = h->step_count;
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_get_dof(tgnh_handle h, double* dof, double* nkt) {
    CHECK_H(h);
    if (dof) std::copy(h->dof.begin(), h->dof.end(), dof);
    if (nkt) std::copy(h->nkbt.begin(), h->nkbt.end(), nkt);
    return TGNH_OK;
}

static bool chain_section(tgnh_handle h, int which, int* off, int* len) {
    switch (which) {
        case 0: *off = h->L.off_eta; *len = h->L.len_eta; return true;
        case 1: *off = h->L.off_etaDot; *len = h->L.len_etaDot; return true;
        case 2: *off = h->L.off_etaDotDot; *len = h->L.len_etaDotDot; return true;
        case 3: *off = h->L.off_etaMass; *len = h->L.len_etaMass; return true;
        default: return false;
    }
}
extern "C" tgnh_status tgnh_get_thermostat_len(tgnh_handle h, int which, int* len) {
    CHECK_H(h);
    int off;
    if (!len) return fail(TGNH_ERR_ARG, "null out");
    if (!chain_section(h, which, &off, len)) return fail(TGNH_ERR_ARG, "bad thermostat array id");
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_get_thermostat_state(tgnh_handle h, int which, void* stream, double* out) {
    CHECK_H(h);
    int off, len;
    if (!chain_section(h, which, &off, &len)) return fail(TGNH_ERR_ARG, "bad thermostat array id");
    return read_state(h, off, len, (hipStream_t)stream, out);
}
extern "C" tgnh_status tgnh_set_thermostat_state(tgnh_handle h, int which, void* stream, const double* in) {
    CHECK_H(h);
    int off, len;
    if (!chain_section(h, which, &off, &len)) return fail(TGNH_ERR_ARG, "bad thermostat array id");
    if (!in) return fail(TGNH_ERR_ARG, "null in");
    tgnh_status rc = deferred_guard(h, "tgnh_set_thermostat_state"); if (rc) return rc;
    h->ke_carry = false;
    if (h->host_only) { std::copy(in, in + len, h->h_state.begin() + off); return TGNH_OK; }
    HIP_OK(hipSetDevice(h->device));
    rc = materialize_chain(h, (hipStream_t)stream); if (rc) return rc;
    // both copies: the in-kernel chain rewrites only the fields it advances in the staging block, and the next commit
    // copies that block over d_state whole -- a field set here alone (etaMass, an unused etaDot slot) would revert
    HIP_OK(hipMemcpyAsync(h->d_state + off, in, sizeof(double) * len, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_OK(hipMemcpyAsync(h->d_stage + off, in, sizeof(double) * len, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    return TGNH_OK;
}

static const std::vector<int>* topo_vec(tgnh_handle h, int which) {
    switch (which) {
        case 0: return &h->normal;
        case 1: return &h->pair_drude;
        case 2: return &h->pair_parent;
        case 3: return &h->group;
        case 4: return &h->resid;
        case 5: return &h->res_count;
        case 6: return &h->res_first;
        case 7: return &h->tile_start;
        default: return nullptr;
    }
}
extern "C" tgnh_status tgnh_get_topology_len(tgnh_handle h, int which, int* len) {
    CHECK_H(h);
    if (!len) return fail(TGNH_ERR_ARG, "null out");
    if (which == 8) { *len = (int)h->meta.size(); return TGNH_OK; }
    if (which == 9) { *len = 2 * (int)h->wave_tile.size(); return TGNH_OK; }      // wave tiles: (first slot, largest molecule) pairs, one more than tiles; 0 = none
    if (which == 10) { *len = (int)h->wmeta.size(); return TGNH_OK; }
    if (which == 11) { *len = (int)h->tile_pat.size(); return TGNH_OK; }        // per 512-slot tile: period | molecules per period << 8 | pattern << 16 (0: per-slot words)
    if (which == 12) { *len = (int)h->wtile_pat.size(); return TGNH_OK; }       // per wave tile: period | pattern << 8
    if (which == 13) { *len = (int)h->pattern.size(); return TGNH_OK; }         // the patterns, 64 words each
    if (which == 14) { *len = (int)h->wpattern.size(); return TGNH_OK; }
    const std::vector<int>* v = topo_vec(h, which);
    if (!v) return fail(TGNH_ERR_ARG, "bad topology array id");
    *len = (int)v->size();
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_get_topology(tgnh_handle h, int which, int32_t* out) {
    CHECK_H(h);
    if (!out) return fail(TGNH_ERR_ARG, "null out");
    auto put = [&](const void* src, size_t bytes) { if (bytes) std::memcpy(out, src, bytes); return TGNH_OK; };     // (an empty array has no data())
    if (which == 8) return put(h->meta.data(), sizeof(uint32_t) * h->meta.size());
    if (which == 9) return put(h->wave_tile.data(), sizeof(int2) * h->wave_tile.size());
    if (which == 10) return put(h->wmeta.data(), sizeof(uint32_t) * h->wmeta.size());
    if (which == 11) return put(h->tile_pat.data(), sizeof(uint32_t) * h->tile_pat.size());
    if (which == 12) return put(h->wtile_pat.data(), sizeof(uint32_t) * h->wtile_pat.size());
    if (which == 13) return put(h->pattern.data(), sizeof(uint32_t) * h->pattern.size());
    if (which == 14) return put(h->wpattern.data(), sizeof(uint32_t) * h->wpattern.size());
    const std::vector<int>* v = topo_vec(h, which);
    if (!v) return fail(TGNH_ERR_ARG, "bad topology array id");
    std::copy(v->begin(), v->end(), out);
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_compute_kinetic_energies(tgnh_handle h, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = settle_kick(h, s); if (rc) return rc;
    rc = materialize_chain(h, s); if (rc) return rc;      // ke_red is about to be overwritten
    const int dir = h->sweep_reverse;                      // a query leaves the sweep direction as it found it: the step's next
    rc = run_tile(h, OP_KE, KID_KE, s);                    // KE launch then sums in the same order, to the same bits
    h->sweep_reverse = dir;
    if (rc) return rc;
    if (h->gather_chain) {
        rc = run_chain_gather(h, s, true); if (rc) return rc;
    } else {
        ChainArgs a = chain_args(h);
        a.do_sum = 1; a.do_chain = 0;
        if (h->xchg_on) { a.x_send = 1; a.x_wait = 1; }
        if (!h->tail_summed) HIP_OK(launch_chain(a, s));       // (summed by the KE launch itself where the step's own KE launch is: the same bits)
        h->tail_summed = false;
        if (!h->xchg_on && h->allreduce && h->allreduce(h->d_state + h->L.off_ke_red, h->L.NT, (void*)s, h->allreduce_user) != 0)
            return fail(TGNH_ERR_HIP, "all-reduce hook failed");
    }
    HIP_OK(hipMemcpyAsync(h->d_state + h->L.off_ke, h->d_state + h->L.off_ke_red, sizeof(double) * h->L.NT,
                          hipMemcpyDeviceToDevice, s));
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// harness
// ---------------------------------------------------------------------------
extern "C" tgnh_status tgnh_harness_force(tgnh_handle h, const void* x0, double k_drude, double k_tether,
                                          void* force_out, void* stream) {
    tgnh_status rc = entry(h, true); if (rc) return rc;
    if (!force_out) return fail(TGNH_ERR_ARG, "null force_out");
    if (h->generic) {                                      // the gather path: partners by index (no per-slot word to read an offset from)
        if (!x0) x0 = h->d_g_x0;
        if (!x0) return fail(TGNH_ERR_ARG, "null x0 and no packed sites (tgnh_harness_pack_sites)");
        Timed t(h, (hipStream_t)stream, KID_FORCE);
        HIP_OK(launch_gather_force(h->d.precision, gather_args(h, nullptr), x0, reinterpret_cast<long long*>(force_out), k_drude, k_tether, (hipStream_t)stream));
        return TGNH_OK;
    }
    if (!x0 && !h->d_sflag && !h->lat_on) return fail(TGNH_ERR_ARG, "null x0 and no packed sites (tgnh_harness_pack_sites)");
    ForceArgs a{};
    a.posq = h->posq; a.posq_corr = h->posq_corr; a.x0 = x0; a.meta = h->d_meta;
    if (!x0 && h->lat_on) {
        a.lat_k = h->lat_k; a.lat_side = h->lat_side; a.lat_mol0 = h->lat_mol0; a.lat_spacing = h->lat_spacing; a.lat_tab = h->d_lat_tab;
        a.lat_inv_k = 1.0 / h->lat_k; a.lat_inv_side = 1.0 / h->lat_side; a.lat_inv_side2 = 1.0 / ((double)h->lat_side * h->lat_side);
    } else if (!x0) { a.sflag = h->d_sflag; a.sbase = h->d_sbase; a.sites = h->d_sites; }
    a.force = reinterpret_cast<long long*>(force_out);
    a.n = h->d.num_particles; a.padded = h->d.padded_num_particles;
    a.k_drude = k_drude; a.k_tether = k_tether;
    a.reverse = h->sweep_reverse;
    if (h->alternate_sweeps) h->sweep_reverse ^= 1;
    Timed t(h, (hipStream_t)stream, KID_FORCE);
    HIP_OK(launch_force(h->d.precision, a, (hipStream_t)stream));
    return TGNH_OK;
}

// The sites the force kernel needs, without what it does not: x0 spends 16-32 B on every slot for a site only the tethered
// ones have (3 of 5 in SWM4 water), and the meta word 4 B for two bits and a small offset.  Host-side, once.
extern "C" tgnh_status tgnh_harness_pack_sites(tgnh_handle h, const void* x0) {
    tgnh_status rc = entry(h, false); if (rc) return rc;
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle");
    if (!x0) return fail(TGNH_ERR_ARG, "null x0");
    const int N = h->d.num_particles;
    const size_t rs = h->d.precision == TGNH_PREC_DOUBLE ? sizeof(double) : sizeof(float);
    if (h->generic) {                                      // the gather path keeps the sites as they came (its force kernel reads them by index)
        h->lat_on = false;
        if (!h->d_g_x0) HIP_OK(hipMalloc(&h->d_g_x0, (size_t)std::max(N, 1) * 4 * rs));
        HIP_OK(hipMemcpy(h->d_g_x0, x0, (size_t)N * 4 * rs, hipMemcpyDeviceToDevice));
        return TGNH_OK;
    }
    std::vector<unsigned char> raw((size_t)std::max(N, 1) * 4 * rs);
    HIP_OK(hipMemcpy(raw.data(), x0, (size_t)N * 4 * rs, hipMemcpyDeviceToHost));
    std::vector<uint8_t> flag((size_t)std::max(N, 1), 0);
    std::vector<uint32_t> base((size_t)(N + 63) / 64 + 1, 0);
    std::vector<unsigned char> sites;
    sites.reserve((size_t)N * 3 * rs);
    uint32_t count = 0;
    for (int i = 0; i < N; i++) {
        if ((i & 63) == 0) base[i >> 6] = count;
        const unsigned char* rec = raw.data() + (size_t)i * 4 * rs;
        const bool tethered = rs == sizeof(double) ? reinterpret_cast<const double*>(rec)[3] != 0.0 : reinterpret_cast<const float*>(rec)[3] != 0.0f;
        const uint32_t m = h->meta[i], role = m & 3u;
        const int off = (int)((m >> 10) & 2047u) - 1024;
        const uint32_t o5 = role == ROLE_NORMAL ? 16u : (off >= -15 && off <= 15 ? (uint32_t)(off + 16) : 0u);
        flag[i] = (uint8_t)(role | (tethered && role != ROLE_DRUDE ? 4u : 0u) | (o5 << 3));
        if (flag[i] & 4u) { sites.insert(sites.end(), rec, rec + 3 * rs); count++; }
    }
    if (h->d_sflag) (void)hipFree(h->d_sflag);
    if (h->d_sbase) (void)hipFree(h->d_sbase);
    if (h->d_sites) (void)hipFree(h->d_sites);
    h->d_sflag = nullptr; h->d_sbase = nullptr; h->d_sites = nullptr;
    h->lat_on = false;
    if (h->lat_k > 0) {
        // The hint (tgnh_harness_lattice_hint) is taken only if it reproduces what was just packed: every slot's byte is molecule
        // 0's, every tethered site is fl(fl64(lattice index x spacing) + geom) in the position type -- the arithmetic of the kernel
        const int k = h->lat_k, side = h->lat_side;
        bool ok = N > 0 && N % k == 0 && (long long)side * side * side >= (long long)h->lat_mol0 + N / k;
        for (int i = 0; i < N && ok; i++) {
            const int pos = i % k, mol = h->lat_mol0 + i / k;
            ok = flag[i] == flag[pos] && (flag[i] >> 3) != 0;         // (a partner more than 15 slots away reads the meta word: not here)
            if (ok && (flag[i] & 4u)) {
                const int idx[3] = {mol / (side * side), (mol / side) % side, mol % side};
                const unsigned char* rec = raw.data() + (size_t)i * 4 * rs;
                for (int ax = 0; ax < 3 && ok; ax++) {
                    volatile double c = (double)idx[ax] * h->lat_spacing;          // (two roundings, as numpy's: no contraction)
                    const double v = c + h->lat_geom[(size_t)pos * 3 + ax];
                    ok = rs == sizeof(double) ? reinterpret_cast<const double*>(rec)[ax] == v : reinterpret_cast<const float*>(rec)[ax] == (float)v;
                }
            }
        }
        if (ok) {
            std::vector<unsigned char> tab(LAT_TAB_BYTES, 0);
            std::memcpy(tab.data(), flag.data(), (size_t)k);
            std::memcpy(tab.data() + 64, h->lat_geom.data(), sizeof(double) * 3 * (size_t)k);
            if (!h->d_lat_tab) HIP_OK(hipMalloc(&h->d_lat_tab, LAT_TAB_BYTES));
            HIP_OK(hipMemcpy(h->d_lat_tab, tab.data(), LAT_TAB_BYTES, hipMemcpyHostToDevice));
            h->lat_on = true;
            return TGNH_OK;                                            // nothing per slot is kept
        }
    }
    HIP_OK(hipMalloc(&h->d_sflag, flag.size()));
    HIP_OK(hipMalloc(&h->d_sbase, base.size() * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&h->d_sites, std::max(sites.size(), (size_t)16) + 16));    // (+16: a 12-byte record may be fetched as four dwords)
    HIP_OK(hipMemcpy(h->d_sflag, flag.data(), flag.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->d_sbase, base.data(), base.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!sites.empty()) HIP_OK(hipMemcpy(h->d_sites, sites.data(), sites.size(), hipMemcpyHostToDevice));
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_harness_lattice_hint(tgnh_handle h, int mol_slots, int side, double spacing, const double* geom, int first_molecule) {
    CHECK_H(h);
    // a new (or withdrawn) hint is unverified until tgnh_harness_pack_sites has checked it slot by slot: the lattice kernel is
    // not taken with it (tgnh_harness_force with x0 = NULL uses the sites packed before, or fails if there are none)
    h->lat_on = false;
    if (mol_slots == 0) { h->lat_k = 0; h->lat_geom.clear(); return TGNH_OK; }
    if (mol_slots < 1 || mol_slots > 64 || side < 1 || side > 1290 || !(spacing > 0) || !geom || first_molecule < 0) return fail(TGNH_ERR_ARG, "bad lattice hint");
    h->lat_k = mol_slots; h->lat_side = side; h->lat_spacing = spacing; h->lat_mol0 = first_molecule;
    h->lat_geom.assign(geom, geom + 3 * (size_t)mol_slots);
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_harness_sites_kind(tgnh_handle h, int* kind) {
    CHECK_H(h);
    if (!kind) return fail(TGNH_ERR_ARG, "null out");
    *kind = h->lat_on ? 2 : (h->d_sflag ? 1 : 0);
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_run_harness(tgnh_handle h, const void* x0, double k_drude, double k_tether,
                                        int nsteps, void* stream) {
    CHECK_H(h);
    for (int i = 0; i < nsteps; i++) {
        tgnh_status rc = tgnh_step_begin(h, stream); if (rc) return rc;
        rc = tgnh_harness_force(h, x0, k_drude, k_tether, const_cast<void*>(h->force), stream); if (rc) return rc;   // Cu :380 call-out
        rc = tgnh_step_end(h, stream); if (rc) return rc;
    }
    return TGNH_OK;
}

extern "C" tgnh_status tgnh_run_steps(tgnh_handle h, int nsteps, void* stream) {
    CHECK_H(h);
    for (int i = 0; i < nsteps; i++) {
        tgnh_status rc = tgnh_step_begin(h, stream); if (rc) return rc;
        rc = tgnh_step_end(h, stream); if (rc) return rc;          // (the force buffer is the caller's business: no call-out here)
    }
    return TGNH_OK;
}

// ---------------------------------------------------------------------------
// timing / roofline bookkeeping
// ---------------------------------------------------------------------------
static void drain_events(tgnh_handle h) {
    for (size_t i = 0; i < h->ev_used; i++) {
        auto& e = h->ev_pool[i];
        if (hipEventSynchronize(e.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { h->t_total[e.kid] += ms; h->t_count[e.kid] += 1; }
    }
    h->ev_used = 0;
}
extern "C" tgnh_status tgnh_timing_enable(tgnh_handle h, int on) {
    CHECK_H(h);
    if (h->host_only) return fail(TGNH_ERR_STATE, "host-only handle");
    HIP_OK(hipSetDevice(h->device));
    if (!on) drain_events(h);
    else {
        for (int k = 0; k < KID_COUNT; k++) { h->t_total[k] = 0; h->t_count[k] = 0; }
        h->ev_used = 0;
        while (h->ev_pool.size() < 2048) {        // created here, not lazily inside somebody's timed region
            tgnh_context::Ev e; e.kid = 0;
            if (hipEventCreate(&e.a) != hipSuccess) break;
            if (hipEventCreate(&e.b) != hipSuccess) { (void)hipEventDestroy(e.a); break; }
            h->ev_pool.push_back(e);
        }
    }
    h->timing = on != 0;
    h->timing_only = on >= 2 ? on - 2 : -1;   // on = 2 + kernel id: time that kernel only (2 events per step, not 8)
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_timing_read(tgnh_handle h, int kernel, double* total_ms, int64_t* launches) {
    CHECK_H(h);
    if (kernel < 0 || kernel >= KID_COUNT) return fail(TGNH_ERR_ARG, "bad kernel id");
    if (!h->host_only) { HIP_OK(hipSetDevice(h->device)); drain_events(h); }
    if (total_ms) *total_ms = h->t_total[kernel];
    if (launches) *launches = h->t_count[kernel];
    return TGNH_OK;
}
extern "C" tgnh_status tgnh_algorithmic_bytes(tgnh_handle h, int kernel, double* bytes) {
    CHECK_H(h);
    if (!bytes) return fail(TGNH_ERR_ARG, "null out");
    // state arrays only.  V = velocity vec4, F = 3 x int64, X = position (+correction) per direction.
    const double N = h->d.num_particles;
    const double V = h->d.precision == TGNH_PREC_SINGLE ? 16 : 32;
    const double F = 24;
    const double X = h->d.precision == TGNH_PREC_SINGLE ? 16 : 32;   // mixed: 16 posq + 16 correction; double: 32
    double b = 0;
    switch (kernel) {
        case KID_SKD: b = N * (2 * V + F + 2 * X); break;       // scale+kick+drift: V r/w, F r, X r/w
        case KID_KICK_KE: b = N * (V + F); break;               // kick+KE: V r, F r -- the kicked velocities feed the sums only (every fused structure since round 4)
        case KID_SCALE: b = N * (2 * V + (h->end_folded ? F : 0)); break;    // rescale: V r/w (+ F r where it forms the kicked velocities again)
        case KID_KE: b = N * V; break;                          // KE: V r
        case KID_FORCE: b = N * (X + F); break;                 // harness: X r, F w (x0 excluded)
        case KID_STEP:       // step_kernel: its two passes.  Deferred: (V r, F r) + (V r/w, F r, X r/w); the reference's pass
                             // structure: begin (V r) + (V r/w, F r, X r/w) and end (V r, F r) + (V r/w, F r), averaged per launch
            b = (h->d.flags & TGNH_FLAG_DEFER_SCALE) ? N * (3 * V + 2 * F + 2 * X) : N * (6 * V + 3 * F + 2 * X) / 2; break;
        default: b = 0;
    }
    *bytes = b;
    return TGNH_OK;
}
