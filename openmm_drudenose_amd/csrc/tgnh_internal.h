// tgnh_internal.h -- shared between the host side (tgnh_host.cpp) and the
// gfx950 kernels (tgnh_kernels.hip).  Not part of the ABI.
#ifndef TGNH_INTERNAL_H_
#define TGNH_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/drude_tgnh.h"

namespace tgnh {

// ---- tile geometry ---------------------------------------------------------
// One 256-thread work-group (4 wavefronts of 64) owns a tile of up to TILE_SLOTS
// consecutive particle slots whose ends never cut a Drude pair or (TGNH+COM) a
// residue, so partner and molecular-COM look-ups stay inside the tile's LDS image.
#ifndef TGNH_SPT
#define TGNH_SPT 2
#endif
#ifndef TGNH_TBLOCK
#define TGNH_TBLOCK 256
#endif
constexpr int BLOCK = 256;                  // chain / force / plain-KE kernels
constexpr int TBLOCK = TGNH_TBLOCK;          // tile kernel work-group size
constexpr int SPT = TGNH_SPT;                // slots per thread
constexpr int TILE_SLOTS = TBLOCK * SPT;      // 512 at SPT = 2
constexpr int TILE_RES = TILE_SLOTS / 2;     // residues per tile the LDS COM table holds
constexpr int GRID_CAP = 2048;               // upper bound of the persistent grid (256 CUs x 8 work-groups)
constexpr int MAX_GROUPS = 32;               // temperature groups; <= 8: per-lane register KE bins (GB template 1/4/8), else per-wave LDS bins (GB 0)

// ---- packed per-slot topology word ----------------------------------------
//  bits  0..1  role: 0 normal, 1 Drude particle (pair.x), 2 parent (pair.y)
//  bits  2..9  temperature group
//  bits 10..20 partner offset + 1024 (partner slot = slot + offset)
//  bits 21..31 residue index local to the tile
constexpr uint32_t ROLE_NORMAL = 0, ROLE_DRUDE = 1, ROLE_PARENT = 2;
inline uint32_t pack_meta(uint32_t role, uint32_t group, int partner_off, uint32_t local_res) {
    return role | (group << 2) | ((uint32_t)(partner_off + 1024) << 10) | (local_res << 21);
}

// ---- wave tiles: the kinetic-energy passes without barriers ---------------------------------------------------------
// A KE pass (KE, kick+KE) only needs, per slot, its Drude partner and its molecule's centre-of-mass velocity.  tile_kernel
// gets both from an LDS image of a 512-slot tile shared by four wavefronts: store, barrier, ONE thread per molecule walks
// its slots (12 of 64 lanes busy, dependent reads), barrier, look-ups, barrier, and only then the work-group's next loads.
// When every molecule fits a wavefront the slot range is also cut into WAVE tiles of <= 64 consecutive slots (never through a
// pair or a molecule), one per wavefront, with a wavefront-PRIVATE LDS image: the wavefront stores its 64 velocities and
// masses (4 conflict-free ds_write), and every lane then sums its own molecule from the image -- all 64 lanes busy, plain
// ds_read_b64 (2 LDS cycles per wavefront instruction; shuffling through the crossbar instead, 38 ds_bpermute per tile, or
// through the vector ALU's row moves, ~110 DPP moves per tile, both measured slower than the LDS pipe it was meant to spare:
// profiles/r03_ke_pass.md).  A wavefront's own LDS operations are processed in order, so there is no barrier anywhere.
// Its per-slot word:
//  bits  0..1  role            bits 2..9  temperature group       bits 10..16 partner offset + 64
//  bits 17..22 position of the slot inside its molecule            bits 23..28 slots of the molecule - 1
constexpr int WAVE_SLOTS = 64;
constexpr int WBLOCK = 512;                 // wstep_kernel's work-group: 8 wavefronts hand in ONE row of sums (half the rows to collect)
inline uint32_t pack_wmeta(uint32_t role, uint32_t group, int partner_off, uint32_t pos_in_mol, uint32_t mol_slots_m1) {
    return role | (group << 2) | ((uint32_t)(partner_off + 64) << 10) | (pos_in_mol << 17) | (mol_slots_m1 << 23);
}

// ---- tiles of identical molecules ------------------------------------------------------------------------------------
// A box is mostly one kind of small molecule (water), and for a tile that holds nothing else the per-slot word is a function
// of the slot's position in the tile: word(k) = pattern[k mod P] (+ (k div P) * molecules per period << 21 for the 512-slot
// tiles' molecule index), P = the molecule's slots or a few molecules' (every tenth water tagged: ten).  Such a tile is marked at create and its kernels read the P words of the pattern (a few cache
// lines for the whole launch) instead of 4 B per slot from HBM: 20 MB per launch at 5 M slots.
constexpr int PATTERN_WORDS = 64;       // longest period

// ---- operation mask of the tile kernel --------------------------------------
enum : int {
    OP_SCALE = 1,      // A6  velocity rescale          (K integrateDrudeTGNHChain)
    OP_KICK = 2,       // A7  half kick                 (K integrateDrudeTGNHVelocities)
    OP_DRIFT = 4,      // A8  x += dt v (no constraints)
    OP_KE = 8,         // A3/A4 kinetic-energy partial sums
    OP_POSDELTA = 16,  // write posDelta = dt v         (constrained path)
    OP_MOVE = 32,      // x += posDelta, v = posDelta/dt (K integrateDrudeTGNHPositions)
    OP_PREKICK = 64,   // first apply the half kick the previous step's kick+KE launch left unstored (DEFER_SCALE)
    OP_NOSTORE = 128   // kick+KE: the kicked velocities feed the sums only; the next launch redoes the kick (OP_PREKICK)
};

// kernel ids for timing / algorithmic bytes (include/drude_tgnh.h)
enum : int { KID_SKD = 0, KID_KICK_KE = 1, KID_SCALE = 2, KID_KE = 3, KID_CHAIN = 4, KID_FORCE = 5,
             KID_OTHER = 6, KID_STEP = 7, KID_COUNT = 8 };

// thermostat block in device memory (doubles), offsets in doubles
struct ChainLayout {
    int mode, G, NT, C, use_drude_chains;
    int len_eta, len_etaDot, len_etaDotDot, len_etaMass;
    int off_eta, off_etaDot, off_etaDotDot, off_etaMass;
    int off_nkbt;      // [NT]
    int off_ke;        // [NT] last KE (before the chain)
    int off_ke_red;    // [NT] summed KE buffer (all-reduced in place when sharded)
    int off_scale;     // [NT] factors the next rescale applies (= scale_a * scale_b)
    int off_scale_a;   // [NT] factors of the chain call itself
    int off_scale_b;   // [NT] DEFER_SCALE: factors of the pre-run first half of the next step (else 1)
    int off_kesum;     // [1]
    int off_ke_post;   // [NT] KE after the chain (KE * prod exp)
    int total;
    // dualNH bookkeeping (Ref :139-154)
    int numTempGroup, idxMaxNHChains, iNumNHChains;
    // one-link chains (Chain1Map): thermostat lane itg keeps link 0 at (itg >> c1_shift) * c1_mul of etaDot, its dummy
    // c1_add further on, eta / etaDotDot / etaMass at itg >> c1_shift; lane c1_unused carries no thermostat; lanes
    // below c1_guard_below have the etaMass > 0 guard.  TGNH: 0, 2, 1, -1, NT-1.  dualNH (useDrudeNHChains): 1, 1, 2, 1, 0.
    int c1_shift, c1_mul, c1_add, c1_unused, c1_guard_below;
    int c1_quirk;      // dualNH without useDrudeNHChains: the real chain is damped by the Drude thermostat (chain1q_run)
};

// Mailbox exchange of the per-thermostat kinetic-energy sums between the ranks of a sharded run (one GPU each),
// the all-reduce of SURVEY 8e done with plain stores over xGMI instead of a collective launch:
//   mailbox (uncached device memory, one per rank, mapped into every peer by IPC):
//       cell[parity][source rank][thermostat] = 16 bytes: two words { 32 bits of the double, 32-bit sequence number }
//   send  (chain_kernel after the row sum): seq = ++*counter; every value goes into cell[seq & 1][my rank][i] of EVERY
//         rank's mailbox (its own too).  Data and sequence number travel in the same 8-byte store, which is atomic,
//         so there is no flag to order behind the data and no fence on either side;
//   wait  (the chain wavefront of every work-group of the next rescale launch, or chain_kernel): read the cells of
//         the own mailbox until all world x NT of them carry seq, add them in rank order -> same bits on every rank.
//         Both parities and the counter are fetched in one batch: one memory round trip when the data is already there.
// Two parities: a rank can be at most one exchange ahead of a peer that is still reading.  Every spin is bounded; a
// time-out sets status bit 2 and the `dead` latch (later waits return at once), so a broken link ends in an error
// report, never in a hung device.
constexpr int XCHG_NT_PAD = 40;             // cells per source rank (NT <= 34)
constexpr int XCHG_CELL_U64 = 2;            // 8-byte words per cell
constexpr int XCHG_MAX_WORLD = 16;
constexpr unsigned XCHG_SPIN_LIMIT = 1000000u;  // polls (each a round trip to memory, ~2 us) before giving up: seconds
constexpr unsigned CENSUS_SPIN_LIMIT = 20000u;  // the residency census of step_kernel: tens of milliseconds
// Replicas: a mailbox holds XCHG_REPLICAS copies of its cells, a sender stores into all of them and waiting work-group b
// polls copy b % XCHG_REPLICAS.  One copy is one 48-byte neighbourhood, i.e. ONE memory channel that all 768 waiting
// work-groups of a launch poll at once (their requests queue there, and the poll that finally sees the data waits behind
// everybody else's); the copies lie a page and a channel apart (stride = k x 4096 + 256 bytes).
#ifndef TGNH_XCHG_REPLICAS
#define TGNH_XCHG_REPLICAS 32
#endif
constexpr int XCHG_REPLICAS = TGNH_XCHG_REPLICAS;
// (from one copy to the next, in 8-byte words: room for the largest world, so that the stride is a compile-time constant)
constexpr size_t XCHG_REPLICA_U64 =
    ((sizeof(unsigned long long) * 2 * (size_t)XCHG_MAX_WORLD * XCHG_NT_PAD * XCHG_CELL_U64 + 4095) / 4096 * 4096 + 256) / sizeof(unsigned long long);
constexpr size_t XCHG_MAILBOX_BYTES(int /*world*/) { return sizeof(unsigned long long) * XCHG_REPLICA_U64 * XCHG_REPLICAS; }

struct XchgArgs {
    int on;                     // 0 off
    int world, rank;
    unsigned long long* const* peers;   // [world] every rank's mailbox as mapped here (peers[rank] = mine)
    unsigned long long* mine;
    unsigned long long* seq;    // this rank's exchange counter (device)
    unsigned int* dead;         // latch: an exchange timed out
    unsigned int* status;       // the handle's status word (bit 2: exchange time-out)
    unsigned long long* stat;   // [3] wait statistics of work-group 0: sum of ticks, largest, exchanges (wall_clock64, 10 ns); may be null
};

constexpr int CHAIN_INLINE_SUM_NT = 10;     // thermostats (G <= 8) ...
constexpr int CHAIN_INLINE_SUM_ROWS = 256;  // ... and partial rows up to which a rescale launch sums the rows itself

struct ChainArgs {
    ChainLayout L;
    XchgArgs x;
    int x_send, x_wait;        // this launch sends its sums after the row sum / waits for everybody's before the chain
    double* st;                // thermostat block
    uint32_t* status;          // the handle's status word: bit 4 when a chain is handed a NaN kinetic energy (a tail sum that gave up, here
                               // or -- through an all-reduce -- on a peer rank: every rank then reports the failure itself)
    const double* partials;    // [nparts][NT] rows of the tile work-groups, then [nbig][NT] rows at GRID_CAP
    int nparts;
    int nbig;
    const double* stage;       // commit != 0: copy this block over st first (an in-kernel chain left it there)
    int commit;
    int do_sum;                // sum partials -> ke_red
    int do_chain;              // run the chain from ke_red
    int chain_twice;           // DEFER_SCALE: second half of step n and first half of step n+1 back to back
    int lanes;                 // chains of 5-16 links (TGNH): a link per lane (chain_lanes_run) instead of LDS-resident links
    int ke_carry;              // the chain's kinetic energies are the last chain's ke_post (= s^2 KE: the bins of the velocities that
                               // chain's rescale left, TGNH_FLAG_TRUST_STATE_CHANGED) -- no KE pass, no row sum ran for this half step
    double dt;
    int S;
    double dtc, inv_dtc;       // dt / S and its reciprocal, formed on the host (a division is a dozen fp64 instructions of the chain wavefront)
    double realkbT, drudekbT;
};

struct TileArgs {
    void* posq;
    void* posq_corr;
    void* velm;
    const long long* force;
    void* pos_delta;
    const uint32_t* meta;
    const int* tile_start;     // [num_tiles+1]
    const int* tile_res;       // [num_tiles+1] first residue of each tile
    const int2* res_table;     // per-tile molecule entries (count, first slot); count < 0: big molecule -count-1
    const void* big_com;       // mixed4 [num_big] COM velocity (w = 1/M) of the molecules longer than a tile
    const double* scale;       // [NT] velocity scale factors (device)
    double* partials;          // [grid][NT] per-work-group KE partial sums
    uint32_t* status;          // bit0: Drude beyond 2x hard wall; bit2: exchange time-out; bit3: step_kernel's meeting timed out
    unsigned int* sync;        // step_kernel: [1] number of the last launch (its rows' tag); census: [2] work-groups checked in, [3] one gave up
    int census;                // step_kernel: residency check only (tgnh_create)
    unsigned long long* rows;  // step_kernel: [grid][NT] tagged cells (2 words each), uncached
    // wave tiles (wke_kernel)
    const int2* wave_tile;     // [num_wtiles + 1]: (first slot, slots of the tile's largest molecule); the next entry's first slot ends it
    const uint32_t* wmeta;     // per-slot word of the wave tiles (pack_wmeta)
    // Tiles of identical molecules carry no per-slot words (PATTERN_* below): tile_pat[t] = period | molecules per period << 8 |
    // pattern << 16 (0: read meta); the wave tiles' period | pattern << 8 in the high bits of wave_tile[t].y; the patterns, 64 words each
    const uint32_t* tile_pat;
    const uint32_t* pattern;
    const uint32_t* wpattern;
    int num_wtiles;
    // wke_kernel, tail sum: the rows go out as tagged cells (`rows`, `sync`, as in step_kernel) and work-group 0, when its own
    // tiles are done, collects them in row order and leaves the sums in ke_red -- the launch that only summed the rows is gone
    int tail_sum;
    double* ke_red;            // [NT] the thermostat block's summed kinetic energies
    int num_tiles;
    int reverse;               // walk the tiles last-to-first: start where the previous launch ended (its lines are still in the Infinity Cache)
    int padded;
    int num_groups;            // G (internal layout: NT = G+2, [G]=COM, [G+1]=Drude)
    int use_com;
    int hardwall;
    double dt;
    double max_dist;
    double hw_scale;           // sqrt(kB*T_drude)
    // in-kernel chain (numNHChains == 1): every work-group derives the scale factors from the summed kinetic
    // energies itself; work-group 0 writes the advanced thermostat block to a staging copy (work-groups of this
    // launch may start after work-group 0 has finished, so st_in must stay untouched); the next chain_kernel commits it
    int chain_on;
    int sum_rows;              // few partial rows (a small system): the chain wavefront sums them itself -- no chain launch at all
    // a staged thermostat block to commit (work-group 0, first thing): set on KE launches, which always lie between
    // the rescale launch that staged it and the next one that reads st_in; [skip, skip + n) = the summed KE, newer in dst
    int commit_len, commit_skip, commit_skip_n;
    const double* commit_src;
    double* commit_dst;
    int x_wait;                // the chain wavefront takes the kinetic energies from the mailbox exchange
    const double* st_in;
    double* st_out;
    ChainArgs chain;
};

struct BigComArgs {
    const int2* table;         // [n] (count, first slot) of the molecules longer than a tile
    int n;
    const void* velm;
    const long long* force;
    int padded;
    int kick;                  // 1: COM of the velocities after the half kick about to be applied
    double dt;
    void* big_com;             // mixed4 [n] out
    double* partials;          // [n][NT] rows: M v_com^2 in bin G
    int NT, G;
};

struct ForceArgs {
    const void* posq;
    const void* posq_corr;
    const void* x0;            // real4 [N]: tether site (x,y,z), w = 1 if the slot is tethered else 0
    const uint32_t* meta;
    long long* force;
    int n, padded;
    int reverse;
    double k_drude, k_tether;
    // packed sites (tgnh_harness_pack_sites): one byte per slot -- role | tethered << 2 | (partner offset + 16) << 3, 0 in the
    // upper five bits = "read the meta word" --, the number of tethered slots before each 64-slot chunk, and the sites of the
    // tethered slots only, three reals each.  65 B per slot of a mixed-precision water box instead of 77.
    const uint8_t* sflag;
    const uint32_t* sbase;
    const void* sites;
    // lattice sites (tgnh_harness_lattice_hint, verified by tgnh_harness_pack_sites): one molecule of lat_k slots repeated on a
    // simple cubic lattice; flag byte and site offset per slot of the molecule in lat_tab: bytes [64] then doubles [64][3]
    int lat_k, lat_side, lat_mol0;
    double lat_spacing, lat_inv_k, lat_inv_side, lat_inv_side2;
    const unsigned char* lat_tab;
};
constexpr size_t LAT_TAB_BYTES = 64 + 64 * 3 * sizeof(double);

// The gather path (tgnh_gather.hip): the step by global index, the reference's own lists, for topologies the tiles cannot hold
constexpr int GATHER_KE_ROWS = 1024;        // work-groups (= rows of partial sums) of its kinetic-energy kernel: <= GRID_CAP, chain_kernel sums them
constexpr int GATHER_MAX_NT = 2048;         // thermostats: a row of fp64 bins per wavefront, four wavefronts, 64 KiB of LDS
struct GatherArgs {
    void* posq; void* posq_corr; void* velm; const long long* force; void* pos_delta;
    const int* group;          // [n] temperature group (Cu :117)
    const int* resid;          // [n] residue, as index into res_table / com (Cu :118)
    const int2* res_table;     // [n_res] (count, first particle) (Cu :121-125)
    const int* partner;        // [n] the other member of the particle's pair | is-Drude << 31, -1: in no pair (Ref :124-137)
    void* com;                 // mixed4 [n_res] centre-of-mass velocity, w = 1 / M (K comVelm)
    const double* scale;       // [NT]
    double* partials;          // [rows][NT]
    uint32_t* status;
    int n, padded, n_res, G, NT, use_com, hardwall, ops, kick_com;
    int com_lanes;             // gather_com_kernel: lanes per residue (a power of two <= 64)
    double dt, max_dist, hw_scale;
};

// A launcher reports THIS launch's error: whatever an earlier call left behind (e.g. a stream capture the caller
// abandoned) is read off first.
#define TGNH_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// launchers (tgnh_kernels.hip)
hipError_t launch_tile(int precision, int ops, int gb, const TileArgs& a, int grid, size_t lds, hipStream_t s);
int tile_blocks_per_cu(int precision, int ops, int gb, size_t lds, bool multi = false);   // occupancy of that instantiation (multi: its in-kernel chain has 2-4 links)
// step_kernel; kind: 0 a whole deferred step, 1 / 2 the begin / end half of the reference's pass structure, 3 / 4 the same
// around the constraint call-outs (tgnh_kernels.hip: STEP_*)
hipError_t launch_step(int precision, int gb, int kind, const TileArgs& a, int grid, size_t lds, hipStream_t s);
int step_blocks_per_cu(int precision, int gb, int kind, size_t lds);
int step_kind_ops2(int kind);       // operations of the kind's second pass (its LDS needs)
hipError_t launch_wstep(int precision, int gb, bool multi, const TileArgs& a, int grid, hipStream_t s);   // a whole deferred step over wave tiles
int wstep_blocks_per_cu(int precision, int gb, bool multi);
hipError_t launch_wke(int precision, int ops, int gb, const TileArgs& a, int grid, hipStream_t s);   // KE passes over wave tiles
int wke_blocks_per_cu(int precision, int ops, int gb);
hipError_t launch_chain(const ChainArgs& a, hipStream_t s);
hipError_t launch_big_com(int precision, const BigComArgs& a, hipStream_t s);
hipError_t launch_force(int precision, const ForceArgs& a, hipStream_t s);
// the gather path (tgnh_gather.hip)
hipError_t launch_gather_com(int precision, const GatherArgs& a, hipStream_t s);
int gather_ke_grid(const GatherArgs& a);
hipError_t launch_gather_ke(int precision, const GatherArgs& a, int grid, hipStream_t s);
hipError_t launch_gather_rowsum(const double* partials, int nrows, int NT, double* ke_red, hipStream_t s);
hipError_t launch_gather_chain(const ChainArgs& a, double* scratch, hipStream_t s);
hipError_t launch_gather_update(int precision, const GatherArgs& a, hipStream_t s);
hipError_t launch_gather_force(int precision, const GatherArgs& a, const void* x0, long long* force, double k_drude, double k_tether, hipStream_t s);
constexpr int PLAIN_KE_PARTS = 2048;         // work-group partials of the plain kinetic-energy query
hipError_t launch_plain_ke(int precision, const void* velm, const long long* force, int n, int padded,
                           double time_shift, double* out /*[1 + PLAIN_KE_PARTS] device: out[0] = the result*/, hipStream_t s);
size_t tile_lds_bytes(int precision, int ops, bool hardwall, bool use_com);

}  // namespace tgnh

void tgnh_set_error(const std::string& msg);   // sets what tgnh_last_error() returns (tgnh_host.cpp)

struct tgnh_context {
    tgnh_desc d;                      // scalars only; pointers are nulled after create
    int device = 0;
    bool host_only = false;           // device == -1: topology / dof only, no launches
    std::vector<double> h_state;      // host copy of the initial thermostat block
    // host topology (A1), kept for parity queries
    std::vector<double> mass;
    std::vector<int> pair_drude, pair_parent, group, resid, normal;
    std::vector<int> res_count, res_first;
    std::vector<int> tile_start, tile_res;
    std::vector<int2> res_entries;    // per-tile molecule entries
    std::vector<int> big_first, big_count;   // molecules longer than a tile (COM from big_com_kernel)
    int num_big = 0;
    // the gather path: taken when the tiles cannot hold the topology (generic_reason says why); the reference's index lists, per particle
    bool generic = false;
    bool gather_chain = false;        // ... and its chain too: more than 34 thermostats, or links that do not fit the LDS (gather_chain_kernel)
    std::string generic_reason;
    std::vector<int2> g_res_table;
    std::vector<int> g_resid, g_partner;
    int *d_g_group = nullptr, *d_g_resid = nullptr, *d_g_partner = nullptr;
    int2* d_g_res_table = nullptr;
    int g_com_lanes = 64;
    bool g_com_fresh = false;         // the COM table is of the velocities as they are (set by a KE pass; cleared by the next launch that writes velocities and on entry to every entry point)
    void* d_g_com = nullptr;
    double* d_g_scratch = nullptr;    // chains longer than 4 links of more than 34 thermostats: a row of 4 C + 1 doubles each
    void* d_g_x0 = nullptr;           // harness: the tether sites as tgnh_harness_pack_sites was handed them
    int2* d_big_table = nullptr;
    void* d_big_com = nullptr;
    std::vector<uint32_t> meta;
    std::vector<int2> wave_tile;      // wave tiles (empty: some molecule or pair does not fit a wavefront)
    std::vector<uint32_t> wmeta;
    std::vector<uint32_t> tile_pat, wtile_pat;    // per 512-slot tile: period | molecules << 8 | pattern << 16; per wave tile: period | pattern << 8; 0 = none
    std::vector<uint32_t> pattern, wpattern;      // 64 words per pattern
    uint32_t *d_tile_pat = nullptr, *d_pattern = nullptr, *d_wpattern = nullptr;
    int num_wtiles = 0;
    int2* d_wave_tile = nullptr;
    uint32_t* d_wmeta = nullptr;
    bool tail_summed = false;         // the last KE launch summed its rows itself (wke_kernel's tail sum): no row-sum launch
    bool wave_ke = false;             // the KE passes run over the wave tiles (wke_kernel)
    // dof bookkeeping (A2)
    std::vector<double> local_terms, global_terms;   // per thermostat, before CMM correction
    std::vector<double> dof, nkbt;
    double realkbT = 0, drudekbT = 0;
    tgnh::ChainLayout L{};
    // device
    uint32_t* d_meta = nullptr;
    int* d_tile_start = nullptr;
    int* d_tile_res = nullptr;
    int2* d_res_table = nullptr;
    double* d_partials = nullptr;
    double* d_state = nullptr;        // thermostat block
    double* d_stage = nullptr;        // same layout: where an in-kernel chain leaves the advanced block
    bool stage_pending = false;       // d_stage is newer than d_state; the next chain_kernel launch commits it
    int sweep_reverse = 0;            // direction of the next streaming launch (alternates)
    bool alternate_sweeps = true;
    int inline_sum_rows = tgnh::CHAIN_INLINE_SUM_ROWS;
    bool inline_sum_all = false;      // more rows than that (and < 2 M slots): all four wavefronts of the rescale launch sum them (sum_rows = 2)
    bool sum_pending = false;         // with chain_pending: the partial rows are not summed yet either (the rescale launch does both)
    bool chain_pending = false, chain_pending_twice = false;   // summed KE waits for the next rescale launch to run the chain
    bool carry_pending = false;       // ... and that KE is the last chain's ke_post (ChainArgs::ke_carry), not ke_red
    bool carry_ok = false;            // TGNH_FLAG_TRUST_STATE_CHANGED is in effect for this handle (set, unsharded, no molecule spans two groups)
    bool inline_chain = false;        // numNHChains == 1: the chain runs inside the rescale launch
    uint32_t* d_status = nullptr;
    uint32_t* h_status_seen = nullptr;   // pinned: where read-backs of the status word land (periodic, and at every query)
    int failed_code = 0;                 // sticky: a failure the device reported (note_status); every later entry returns it
    std::string failed;
    double* d_scalar = nullptr;       // plain KE: [0] the result, [1 ..] work-group partials
    // harness call-outs (tgnh_harness.hip)
    int4* d_cl_atoms = nullptr; double* d_cl_dist = nullptr; int num_clusters = 0;
    int4* d_vs_atoms = nullptr; double* d_vs_w = nullptr; int num_sites = 0;
    int grid = 0, num_tiles = 0, gb = 1;
    int num_cus = 256, grid_override = 0, ke_parts = 0;
    std::map<int, int> grid_cache;    // ops (+hard-wall bit) -> persistent grid size
    // bound buffers
    void *posq = nullptr, *posq_corr = nullptr, *velm = nullptr, *pos_delta = nullptr;
    const void* force = nullptr;
    // run state
    bool scale_pending = false;       // DEFER_SCALE: velm lags by scale[]
    bool kick_pending = false;        // DEFER_SCALE: velm also lags by the second half kick (force buffer unchanged since)
    bool end_pending = false;         // RESIDENT_STEP: the whole end half of the last step waits for the next step_begin's launch
    unsigned int* d_sync = nullptr;   // step_kernel's meeting: launch number
    unsigned long long* d_rows = nullptr;   // ... and the tagged rows (uncached)
    unsigned long long* self_box = nullptr;      // RESIDENT_STEP without a sharded exchange: a private one-rank mailbox
    unsigned long long* d_self_misc = nullptr;   // ... its counter, latch and peer table
    tgnh::XchgArgs self_x{};
    int resident_grid[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};   // step_kernel's grid by kind and hard wall
    int resident_share = 1, last_step_kind = 0;
    int wresident_per_cu = 0, wresident_grid = 0;   // the same for wstep_kernel (0: none, or no wave tiles)
    int resident_per_cu = 0;          // work-groups of step_kernel per compute unit that the census at create found resident together (0: none -- the handle steps the DEFER_SCALE way)
    bool first_half_done = false;     // DEFER_SCALE: chain for the coming step's first half already run
    bool end_folded = false;          // the last fused end half left the kick to its rescale launch (OP_PREKICK: algorithmic bytes of KID_SCALE)
    bool ke_carry = false;            // TRUST_STATE_CHANGED: ke_post of the last end half IS the kinetic energy of the stored velocities
    double time = 0;
    int64_t step_count = 0;
    tgnh_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    void* rccl_comm = nullptr;        // ncclComm_t: the library enqueues ncclAllReduce itself (tgnh_rccl_init / tgnh_set_rccl_comm)
    bool rccl_owned = false;
    unsigned long long* d_x_stat = nullptr;   // mailbox wait statistics (XchgArgs::stat)
    uint8_t* d_sflag = nullptr;               // harness: packed tether sites (ForceArgs::sflag / sbase / sites), tgnh_harness_pack_sites
    uint32_t* d_sbase = nullptr;
    void* d_sites = nullptr;
    int lat_k = 0, lat_side = 0, lat_mol0 = 0;              // ... or lattice sites: the hint (lat_k = 0: none), and whether pack_sites found it to hold
    double lat_spacing = 0;
    std::vector<double> lat_geom;
    bool lat_on = false;
    unsigned char* d_lat_tab = nullptr;
    // mailbox exchange (tgnh_exchange_*): replaces the hook when attached
    tgnh::XchgArgs x{};
    bool xchg_on = false, xwait_pending = false;
    int x_world = 0, x_rank = 0;
    unsigned long long* x_mailbox = nullptr;  // mine (uncached device memory)
    unsigned long long** d_x_peers = nullptr; // device table of every rank's mailbox
    std::vector<void*> x_opened;              // peers' mailboxes opened by IPC (to close)
    unsigned long long* d_x_seq = nullptr;
    unsigned int* d_x_dead = nullptr;
    // timing
    bool timing = false;
    int timing_only = -1;             // >= 0: only this kernel id is timed
    struct Ev { hipEvent_t a, b; int kid; };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    double t_total[tgnh::KID_COUNT] = {0};
    int64_t t_count[tgnh::KID_COUNT] = {0};
};

#endif
