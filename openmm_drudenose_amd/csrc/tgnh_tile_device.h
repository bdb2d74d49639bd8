// tgnh_tile_device.h -- device code shared by the streaming kernels' translation units (tgnh_kernels.hip today):
// precision traits, the 64-lane sums, the kinetic-energy bins and their work-group reduction, the tagged rows and the meeting
// of the one-launch step kernels (step_meet), and the per-tile work of the wave-tile step kernels (WaveStep).
// Included by .hip files only.  Reference semantics: see tgnh_kernels.hip.
#ifndef TGNH_TILE_DEVICE_H_
#define TGNH_TILE_DEVICE_H_
#include "tgnh_internal.h"

namespace tgnh {
__device__ __forceinline__ double wave_sum(double v);
#ifdef TGNH_TRACE
// Phase timestamps of the streaming kernels (tuning builds only: tools/trace_probe.py, tools/step_trace.py).  16 slots per
// work-group, constant 100 MHz clock, written by thread 0.  One table per translation unit (static: the readers below see
// their own unit's), read by tgnh_debug_read_trace (tgnh_kernels.hip).
static __device__ unsigned long long g_trace[GRID_CAP * 16];
#define TRACE(slot) do { if (threadIdx.x == 0 && (slot) < 16) g_trace[blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#define TRACE_WAIT() __builtin_amdgcn_s_waitcnt(0)
#define TGNH_TRACE_READERS(read_name, clear_name)                                                            \
    extern "C" int read_name(unsigned long long* out) {                                                      \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnh::g_trace), sizeof(unsigned long long) * tgnh::GRID_CAP * 16); \
    }                                                                                                        \
    extern "C" int clear_name() {                                                                            \
        void* p = nullptr;                                                                                   \
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(tgnh::g_trace)) != hipSuccess) return 1;                      \
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * tgnh::GRID_CAP * 16);                       \
    }
// chain_kernel's own clocks (tools/micro/chain_inside.py): wall_clock64 and clock64 at entry, after the prologue and at exit
static __device__ __attribute__((unused)) unsigned long long g_chain_trace[8];
#define CHAIN_TRACE(slot) do { if (threadIdx.x == 0) { g_chain_trace[2 * (slot)] = wall_clock64(); g_chain_trace[2 * (slot) + 1] = clock64(); } } while (0)
#else
#define TRACE(slot) do {} while (0)
#define TRACE_WAIT() do {} while (0)
#define CHAIN_TRACE(slot) do {} while (0)
#endif
}
#include "tgnh_chain_device.h"

namespace tgnh {

template <int PREC> struct Prec;
template <> struct Prec<TGNH_PREC_SINGLE> { typedef float real; typedef float mixed; typedef float4 real4; typedef float4 mixed4; };
template <> struct Prec<TGNH_PREC_MIXED>  { typedef float real; typedef double mixed; typedef float4 real4; typedef double4 mixed4; };
template <> struct Prec<TGNH_PREC_DOUBLE> { typedef double real; typedef double mixed; typedef double4 real4; typedef double4 mixed4; };

__device__ __forceinline__ float4 mk4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__device__ __forceinline__ double4 mk4(double x, double y, double z, double w) { return make_double4(x, y, z, w); }
__device__ __forceinline__ float rcp_(float x) { return 1.0f / x; }
// fp64 reciprocal of a normal, non-zero number (masses and their sums): hardware seed + two Newton steps, 5
// instructions and <= 1-2 ulp, where the IEEE division is 11 (it also scales denormals and fixes up specials).
__device__ __forceinline__ double rcp_(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }

// Sum over the 64 lanes of a wavefront, the same value (and the same bits) in every lane.  Data-parallel-primitive moves
// inside the vector ALU -- quads, then rows of 16 (row_shr 4, 8), then row broadcasts; the total lands in lane 63 and is read
// back as a scalar -- instead of six __shfl_xor butterflies: a shuffle is two ds_bpermute_b32 through the LDS crossbar per
// double, ~100 cycles of latency per step, and these sums (the kinetic-energy bins at the end of a pass, the rows collected by
// work-group 0) sit on the path every work-group of a launch waits for.  The order of the additions is fixed.
#ifndef TGNH_WAVE_SUM_DPP
#define TGNH_WAVE_SUM_DPP 1
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(const double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);   // lanes without a source: +0.0
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#if TGNH_WAVE_SUM_DPP
    v = dpp_add<0xb1, 0xf>(v);       // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);       // quad_perm:[2,3,0,1]: every lane of a quad holds the quad's sum
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8: lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
#endif
}


// What a work-group carries through a launch: the LDS carve, its KE accumulators, launch constants.
template <int PREC, int GB> struct TileEnv {
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::mixed4 mixed4;
    // fp64 images are kept component-wise (x[], y[], z[], w[]): a 32-byte double4 per lane is a 2-way bank conflict on
    // every ds_read/ds_write_b128 and on the per-molecule walk (SQ_LDS_BANK_CONFLICT was 48 % of the LDS cycles);
    // 8-byte components at lane stride 8 (or 8 x molecule size) are conflict-free.  float4 images stay packed.
    static constexpr bool SOA = sizeof(mixed) == 8;
    static constexpr int GBR = GB > 0 ? GB : 1;
    mixed4* sv;              // [TILE_SLOTS] velocity image
    mixed4* scom;            // [TILE_RES]   molecular COM velocity, w = 1/M
    mixed4* sx;              // [TILE_SLOTS] position image (hard wall only)
    double* s_scale;         // [NT] velocity scale factors of this launch
    double* wbins0;          // more than 8 groups: one row of fp64 bins per wavefront in LDS, behind the images (GB == 0)
    char* smem;
    int tid, G;
    bool use_com;
    mixed dt, fscale, s_com, s_drude;
    double ke_g[GBR], ke_com, ke_drude;
    __device__ __forceinline__ static void st_img(mixed4* img, int i, const mixed4& u) {
        mixed* imgc = reinterpret_cast<mixed*>(img);
        if (SOA) { imgc[i] = u.x; imgc[TILE_SLOTS + i] = u.y; imgc[2 * TILE_SLOTS + i] = u.z; imgc[3 * TILE_SLOTS + i] = u.w; }
        else img[i] = u;
    }
    __device__ __forceinline__ static mixed4 ld_img(const mixed4* img, int i) {
        const mixed* imgc = reinterpret_cast<const mixed*>(img);
        if (SOA) return mk4(imgc[i], imgc[TILE_SLOTS + i], imgc[2 * TILE_SLOTS + i], imgc[3 * TILE_SLOTS + i]);
        return img[i];
    }
    __device__ __forceinline__ void init(const TileArgs& a, char* smem_, double* s_scale_, bool hardwall_lds) {
        smem = smem_; s_scale = s_scale_;
        sv = reinterpret_cast<mixed4*>(smem); scom = sv + TILE_SLOTS; sx = scom + TILE_RES;
        tid = threadIdx.x; G = a.num_groups; use_com = a.use_com != 0;
        dt = (mixed)a.dt;
        fscale = (mixed)(0.5 * a.dt / 4294967296.0);     // Cu :295
        s_com = 1; s_drude = 1;
        wbins0 = reinterpret_cast<double*>(sx + (hardwall_lds ? TILE_SLOTS : 0));
        clear_ke();
    }
    __device__ __forceinline__ void clear_ke() {
#pragma unroll
        for (int b = 0; b < GBR; b++) ke_g[b] = 0.0;
        ke_com = 0.0; ke_drude = 0.0;
        if (GB == 0) { double* w = wbins0 + (tid >> 6) * G; for (int g = tid & 63; g < G; g += 64) w[g] = 0.0; }
    }
};

// step_kernel's tagged rows: word j (two per thermostat) of row r.  Rows come in blocks of 64 -- the rows one wavefront of the
// collecting work-group reads with one load -- and inside a block word-major: the 64 lanes of a load read 512 contiguous
// bytes (one request per 64-byte line instead of one per lane: the collection is bound by the requests a single compute
// unit issues), and the words of a row lie 512 bytes apart, within reach of a load's immediate offset (one address per row).
__device__ __forceinline__ size_t row_word(const int r, const int j) {
    return ((size_t)(r >> 6) * (2 * CHAIN_INLINE_SUM_NT) + j) * 64 + (r & 63);
}

// Work-group reduction of the fp64 KE bins: 64-lane sums (wave_sum), then one LDS hop; one row of `partials` per work-group.
// TAGGED: the row is read by another work-group of the SAME launch (step_kernel): every sum goes out as a cell of two
// 8-byte words {32 bits of the double, tag} into a.rows -- data and "it is there" in one atomic store, as in the mailboxes.
template <int PREC, int GB, bool TAGGED, int NTH = TBLOCK>
__device__ __forceinline__ void ke_reduce(const TileArgs& a, TileEnv<PREC, GB>& e, const unsigned tag = 0u, double* scratch = nullptr) {
    typedef TileEnv<PREC, GB> E;
    const int tid = e.tid, G = e.G;
    double (&ke_g)[E::GBR] = e.ke_g;
    double ke_com = e.ke_com, ke_drude = e.ke_drude;
    // thermostat b of this work-group's row (tagged rows: row_word's layout)
    auto put = [&](const int b, double v) {
        if (TAGGED) {
            unsigned long long* cell = a.rows + row_word((int)blockIdx.x, 2 * b);
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
            __hip_atomic_store(cell, t | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(cell + 64, t | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else a.partials[(size_t)blockIdx.x * (G + 2) + b] = v;
    };
    double* sred = scratch ? scratch : reinterpret_cast<double*>(e.smem);   // [TBLOCK/64][GB+2]
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int b = 0; b < GB; b++) ke_g[b] = wave_sum(ke_g[b]);
    if (GB == 0) __syncthreads();                                // every wave's LDS bins are final
    ke_com = wave_sum(ke_com);
    ke_drude = wave_sum(ke_drude);
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < GB; b++) sred[wv * (GB + 2) + b] = ke_g[b];
        sred[wv * (GB + 2) + GB] = ke_com;
        sred[wv * (GB + 2) + GB + 1] = ke_drude;
    }
    __syncthreads();
    if (tid < GB + 2) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NTH / 64; w++) s += sred[w * (GB + 2) + tid];   // fixed order
        if (tid < GB) { if (tid < G) put(tid, s); }
        else put(G + (tid - GB), s);
    }
    if (GB == 0) {
        const double* w0 = e.wbins0;
        for (int g = tid; g < G; g += NTH) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NTH / 64; w++) s += w0[w * G + g];     // fixed order
            put(g, s);
        }
    }
}

// (mixed)force: the fixed-point force as a floating-point number, rounded once.  For doubles hi 2^32 + lo in one fma --
// both parts are exact, so this is the correctly rounded conversion (the bits of the cast) in 3 instructions instead of 4.
__device__ __forceinline__ double force_as(const long long f, double) {
    return fma((double)(int)(f >> 32), 4294967296.0, (double)(unsigned)f);
}
__device__ __forceinline__ float force_as(const long long f, float) { return (float)f; }

// Wave tiles of identical molecules (PATTERN_WORDS, tgnh_internal.h): a lane's position in the pattern is the same in every such
// tile (they all start on a molecule), so the lane keeps its word in a register and fetches it again only when a tile of another
// pattern comes by -- for a water box once per launch.  pat = period | pattern << 8, 0 = this tile's words are read from wmeta.
struct PatternWord {
    uint32_t pat = 0u, word = 0u;
    __device__ __forceinline__ bool of(const TileArgs& a, const uint32_t p, const int lane) {     // wavefront-uniform
        if ((p & 255u) == 0u) return false;
        if (p != pat) {
            const int period = (int)(p & 255u);
            const int q = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)period));     // lane div period (never within rounding of an integer)
            word = a.wpattern[(p >> 8) * PATTERN_WORDS + (uint32_t)(lane - q * period)];
            pat = p;
        }
        return true;
    }
};

// Work-group 0 collects the tagged rows of a launch (ke_reduce<TAGGED>): thread t owns rows t, t + NTH, ...: it polls their cells
// until all carry `want` and adds them in row order into acc[] (fixed order: reproducible bits).  Returns false when a row
// never came (bounded polling).
template <int GB, bool LEAN, int NTH>
__device__ __forceinline__ bool collect_rows(const TileArgs& a, const int tid, const int grid, const int NT, const unsigned long long want,
                                             double (&acc)[GB + 2]) {
    constexpr int NTM = GB + 2;
    bool ok = true;
    // RB rows of this thread per batch of loads (all of them for the resident grid of 768 when G = 1): once the last row
        // is there, one more round trip sees everything -- polled row after row, a thread whose first row came last paid
        // a round trip for each of the others behind it.  CH thermostats of a row per batch: the registers a batch takes
        // (2 x CH x RB words) do not grow with the number of temperature groups.  A row seen complete is not read again: a
        // round waits for all of its loads together (~1 us with everything polled), and the round that finally sees the last
        // row is a short one when it polls the stragglers only (last row -> all seen 1.6 instead of 2.3 us at 625 k slots).
        constexpr int RB = (GB == 1 && !LEAN) ? 3 : 1, CH = 3;
#pragma unroll 1
        for (int r0 = tid; r0 < grid && ok; r0 += RB * NTH) {
#pragma unroll 1
            for (int b0 = 0; b0 < NT && ok; b0 += CH) {
                unsigned long long w[RB][2 * CH];
                bool have[RB];                              // a row seen complete is not read again: later rounds poll the stragglers only
#pragma unroll
                for (int k = 0; k < RB; k++) have[k] = r0 + k * NTH >= grid;
                unsigned n = 0;
                for (;;) {
#pragma unroll
                    for (int k = 0; k < RB; k++) {
                        if (!have[k]) {
                            const unsigned long long* cell = a.rows + row_word(r0 + k * NTH, 2 * b0);   // the lanes of a load read consecutive words
#pragma unroll
                            for (int b = 0; b < 2 * CH; b++) if (b0 + b / 2 < NT) w[k][b] = xchg_ld(cell + b * 64);
                        }
                    }
                    bool all = true;
#pragma unroll
                    for (int k = 0; k < RB; k++) {
                        if (!have[k]) {
                            bool row = true;
#pragma unroll
                            for (int b = 0; b < 2 * CH; b++) if (b0 + b / 2 < NT) row = row && (w[k][b] >> 32) == want;
                            have[k] = row;
                        }
                        all = all && have[k];
                    }
                    if (all) break;
                    if (++n > XCHG_SPIN_LIMIT) { ok = false; break; }
                }
#pragma unroll
                for (int k = 0; k < RB; k++)                // row order: r0, r0 + 256, ...
#pragma unroll
                    for (int b = 0; b < CH; b++)
                        if (b0 + b < NT && r0 + k * NTH < grid) {
                            const double v = __longlong_as_double((long long)((w[k][2 * b + 1] << 32) | (w[k][2 * b] & 0xffffffffull)));
#pragma unroll
                            for (int t = 0; t < NTM; t++) acc[t] += (t == b0 + b) ? v : 0.0;
                        }
            }
        }
    return ok;
}

// The meeting of step_kernel / wstep_kernel: every work-group hands in its row of kinetic-energy sums (tagged cells), work-group 0
// collects them in a fixed order and sends the sums to the mailbox of every rank, one wavefront of every work-group waits for all
// ranks' sums and runs both chain halves (the scale factors land in sh.s_scale).  `prefetch` is called between handing in the
// row and the wait: the loads the second pass will need.  Returns false when an exchange timed out (nothing may be stored).
struct MeetShared {
    double* s_scale;                                  // [MAX_GROUPS + 2]
    double (*s_part)[CHAIN_INLINE_SUM_NT];            // [TBLOCK / 64]
    double* s_x;                                      // [64 + XCHG_MAX_WORLD * CHAIN_INLINE_SUM_NT]
    int* s_go_p; unsigned* s_gen_p; unsigned long long* s_seq1_p;
    const double* s_block;                            // MULTI: the thermostat block as it was at kernel entry (chains of 2-4 links)
};
template <int PREC, int GB, bool LEAN = false, int NTH = TBLOCK, bool MULTI = false, typename Prefetch>
__device__ __forceinline__ bool step_meet(const TileArgs& a, TileEnv<PREC, GB>& e, const unsigned gen0, const unsigned long long seq0,
                                          Chain1Regs& creg, const MeetShared& sh, Prefetch&& prefetch) {
    double* const s_scale = sh.s_scale; double (*const s_part)[CHAIN_INLINE_SUM_NT] = sh.s_part; double* const s_x = sh.s_x;
    int& s_go = *sh.s_go_p; unsigned& s_gen = *sh.s_gen_p; unsigned long long& s_seq1 = *sh.s_seq1_p;
    const int tid = threadIdx.x, G = a.num_groups, NT = G + 2, grid = (int)gridDim.x;
    const bool chain_wave = tid < 64, leader = blockIdx.x == 0;
    const int itg = tid & 63;
    const ChainLayout& L = a.chain.L;
    // the thermostat block has been read (its values are in registers) before this work-group's row is stored
    if (chain_wave) asm volatile("" :: "v"(creg.eta), "v"(creg.etaDot0), "v"(creg.etaDot1), "v"(creg.etaDotDot), "v"(creg.etaMass), "v"(creg.nkbt) : "memory");
    ke_reduce<PREC, GB, true, NTH>(a, e, gen0 + 1u, s_x);
    TRACE(2);
    // what the second pass needs of the held tile beyond what is in registers (its positions): issued now, needed after the meeting
    prefetch();
    // ... and what the chain can form without the sums (index map, constants, 1/Q, expfac): done while the others still work
    // (single precision: its 16 registers there would cost the kernel its fourth work-group per compute unit)
    constexpr bool EARLY_PRE = PREC != TGNH_PREC_SINGLE && !LEAN;      // (LEAN: wstep_kernel, which lives on a small register count: with it 130 VGPRs, one work-group per compute unit)
    Chain1Pre cpre{};
    if (EARLY_PRE && chain_wave && !L.c1_quirk && !(MULTI && L.C > 1)) cpre = chain1_prepare(a.chain, creg, itg);

    // ---- meet: work-group 0 collects the rows.  Thread t owns rows t, t + 256, ...: it polls their cells until all
    // carry this launch's tag and adds them in row order; then 64-lane sums and one LDS hop, fixed order throughout.
    if (tid == 0) { s_gen = gen0; s_seq1 = seq0 + 1ull; }
    __syncthreads();
    const unsigned long long want = (unsigned long long)(s_gen + 1u);
    if (leader) {
        unsigned long long* const my_peer = xchg_peer_of(a.chain.x, tid);   // for the send: fetched before the collection, not after
        constexpr int NTM = GB + 2;                        // NT = G + 2 <= GB + 2: the register arrays follow the instantiation
        double acc[NTM];
#pragma unroll
        for (int b = 0; b < NTM; b++) acc[b] = 0.0;
        bool ok = true;
        TRACE(6);
        ok = collect_rows<GB, LEAN, NTH>(a, tid, grid, NT, want, acc);
        TRACE(10);
        if (!ok) {                                         // a work-group never handed in its row: nobody goes on (no send below)
            atomicOr(a.status, 8u);
            __hip_atomic_store(a.chain.x.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double* big = a.partials + (size_t)GRID_CAP * NT;               // rows of big_com_kernel (an earlier launch)
        for (int r = tid; r < a.chain.nbig; r += NTH)
#pragma unroll
            for (int b = 0; b < NTM; b++) if (b < NT) acc[b] += big[(size_t)r * NT + b];
#pragma unroll
        for (int b = 0; b < NTM; b++) {
            if (b < NT) {
                const double t = wave_sum(acc[b]);
                if ((tid & 63) == 0) s_part[tid >> 6][b] = t;
            }
        }
        // (the barrier of the hand-over doubles as the vote: one thread that gave up on a row stops the whole send -- incomplete
        // sums under a valid tag would let every waiter, here and on the peer ranks, integrate with wrong scale factors; without
        // the send they time out or see the latch, and nothing is stored)
        const bool all_ok = __syncthreads_and(ok ? 1 : 0) != 0;
        TRACE(7);
        // the send (xchg_send's stores, tgnh_chain_device.h), straight from the four wavefronts' partial sums: every storing
        // thread adds them itself, in wavefront order -- no second hand-over through LDS, no second barrier on this path
        const XchgArgs& x = a.chain.x;
        const unsigned long long seq = s_seq1, stag = (seq & 0xffffffffull) << 32;
        if (tid == 0) { a.sync[1] = s_gen + 1u; *x.seq = seq; }      // the next launch's rows carry the next tag
        const int tpp = NTH / x.world;
        if (all_ok && tid < tpp * x.world) {
            unsigned long long* const base = my_peer + xchg_cell(x, (unsigned)(seq & 1ull), x.rank, 0, 0);
            for (int q = tid / x.world; q < NT * XCHG_REPLICAS; q += tpp) {
                const int copy = q / NT, i = q - copy * NT;
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < NTH / 64; w++) v += s_part[w][i];
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                unsigned long long* cell = base + (size_t)copy * XCHG_REPLICA_U64 + (size_t)i * XCHG_CELL_U64;
                __hip_atomic_store(cell, stag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(cell + 1, stag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        TRACE(13);
    }
    if (chain_wave) {
        bool dead = false;                                 // an exchange has timed out, now or earlier (the latch)
        const double mine = xchg_wait_sum<true>(a.chain.x, NT, itg, s_x + 64, seq0 + 1ull, &dead);
        TRACE(8);
        if (itg == 0) s_go = dead ? 0 : 1;
        if (!dead) {
            const bool write = leader;
            creg.ke = mine;
            if (write && itg < NT) a.st_out[L.off_ke_red + itg] = mine;
            if (write) {                                   // Cu :493-497 (work-group 0 only: the others go straight on to the chain)
                const double kesum = wave_sum(itg < NT ? mine : 0.0);
                if (itg == 63) a.st_out[L.off_kesum] = 0.5 * kesum;
            }
            if (MULTI && L.C > 1) chainN_run<false>(a.chain, sh.s_block, a.st_out, write, s_scale, itg, mine);
            else if (itg < NT) {
                if (L.c1_quirk) chain1q_run(a.chain, creg, a.st_out, write, s_scale, itg);
                else chain1_finish(a.chain, creg, EARLY_PRE ? cpre : chain1_prepare(a.chain, creg, itg), a.st_out, write, s_scale, itg);
            }
        }
    }
    __syncthreads();
    return s_go != 0;
}

template <int PREC> struct WStepIn {
    typename Prec<PREC>::mixed4 v;
    uint32_t meta;
    long long fx, fy, fz;
    typename Prec<PREC>::real4 p;
    float4 c;
    // formed by the first half of the work on a tile (prepare): mass, centre-of-mass velocity of the slot's molecule
    typename Prec<PREC>::mixed mass, cx, cy, cz;
};

// ---------------------------------------------------------------------------
// The per-tile work of the one-launch step over WAVE tiles (wstep_kernel): a wavefront owns <= 64 consecutive
// slots and a private LDS image (velocity x, y, z, mass; position x, y, z for the hard wall); nothing here waits for another
// wavefront.  load_vf / load_x issue a tile's global loads, prepare forms mass, the half kick (KICK), the image and the
// molecule's centre-of-mass velocity and -- ke -- adds the tile's kinetic energies to the bins (pass 1); finish is pass 2:
// rescale, half kick, drift, hard wall, stores.  The arithmetic per slot is tile_body's / wke_kernel's, expression for expression.
// Reference: K :82-113, :138-200 (COM, bins), :249-301 (rescale), :307-365 (kick), :435-466 (drift), :471-574 (hard wall).
// ---------------------------------------------------------------------------
struct WaveBounds { int ws, y, n; };        // first slot; the tile's largest molecule | pattern word << 8; slots

template <int PREC, int GB> struct WaveStep {
    typedef typename Prec<PREC>::real real;
    typedef typename Prec<PREC>::mixed mixed;
    typedef typename Prec<PREC>::real4 real4;
    typedef typename Prec<PREC>::mixed4 mixed4;
    const TileArgs& a;
    TileEnv<PREC, GB>* e;                   // the kinetic-energy bins, in the shape ke_reduce takes them
    const double* s_scale;
    mixed *ix, *iy, *iz, *im, *jx, *jy, *jz;
    mixed4* __restrict__ velm; real4* __restrict__ posq; float4* __restrict__ pcorr;
    int lane, G, nw;
    bool use_com, hardwall;
    mixed dt, fscale;
    PatternWord pw;
    // img: this wavefront's [7][WAVE_SLOTS] LDS image
    __device__ __forceinline__ WaveStep(const TileArgs& a_, TileEnv<PREC, GB>* e_, const double* s_scale_, mixed* img, const int lane_)
        : a(a_), e(e_), s_scale(s_scale_), ix(img), iy(img + WAVE_SLOTS), iz(img + 2 * WAVE_SLOTS), im(img + 3 * WAVE_SLOTS),
          jx(img + 4 * WAVE_SLOTS), jy(img + 5 * WAVE_SLOTS), jz(img + 6 * WAVE_SLOTS),
          velm(reinterpret_cast<mixed4*>(a_.velm)), posq(reinterpret_cast<real4*>(a_.posq)), pcorr(reinterpret_cast<float4*>(a_.posq_corr)),
          lane(lane_), G(a_.num_groups), nw(a_.num_wtiles), use_com(a_.use_com != 0), hardwall(a_.hardwall != 0),
          dt((mixed)a_.dt), fscale((mixed)(0.5 * a_.dt / 4294967296.0)) {}     // Cu :295
    __device__ __forceinline__ static void wfence() {      // a wavefront's LDS operations are processed in order: only the compiler is held
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void bounds(const int ww, WaveBounds& b) const {      // (wavefront-uniform: scalar loads)
        const int2* t = a.wave_tile + (a.reverse ? nw - 1 - ww : ww);
        b.ws = t[0].x; b.y = t[0].y; b.n = t[1].x - b.ws;
    }
    __device__ __forceinline__ void load_vf(const WaveBounds& b, WStepIn<PREC>& in) {          // what pass 1 needs
        const int idx = b.ws + lane;
        const bool patterned = pw.of(a, (uint32_t)b.y >> 8, lane);
        mixed4 v = mk4((mixed)0, (mixed)0, (mixed)0, (mixed)0);
        uint32_t meta = 64u << 10;
        long long fx = 0, fy = 0, fz = 0;
        if (lane < b.n) {
            v = reinterpret_cast<const mixed4*>(a.velm)[idx];
            if (!patterned) meta = a.wmeta[idx];
            fx = a.force[idx]; fy = a.force[idx + a.padded]; fz = a.force[idx + 2 * a.padded];
        }
        if (patterned && lane < b.n) meta = pw.word;
        in.v = v; in.meta = meta; in.fx = fx; in.fy = fy; in.fz = fz;
    }
    __device__ __forceinline__ void load_x(const WaveBounds& b, WStepIn<PREC>& in) const {           // ... and what pass 2 needs on top
        const int idx = b.ws + lane;
        if (lane < b.n) {
            in.p = reinterpret_cast<const real4*>(a.posq)[idx];
            if (PREC == TGNH_PREC_MIXED) in.c = reinterpret_cast<const float4*>(a.posq_corr)[idx];      // K :443-445
        }
    }
    // first half of the work on a tile: mass, the pending half kick, the wavefront's image, the molecule's centre-of-mass
    // velocity; KE: this tile's kinetic energies go into the bins (pass 1)
    template <bool KICK = true>
    __device__ __forceinline__ void prepare(WStepIn<PREC>& t, const WaveBounds& bd, const bool ke) {
        mixed4& v = t.v;
        const uint32_t m = t.meta;
        t.mass = v.w != 0 ? rcp_(v.w) : (mixed)0;
        if (KICK) {                                                      // A7 (Cu :384-388), per particle; w = 0: c = 0, v unchanged
            const mixed c = fscale * v.w;
            v.x += c * force_as(t.fx, (mixed)0);
            v.y += c * force_as(t.fy, (mixed)0);
            v.z += c * force_as(t.fz, (mixed)0);
        }
        ix[lane] = v.x; iy[lane] = v.y; iz[lane] = v.z; im[lane] = t.mass;
        wfence();
        mixed cx = 0, cy = 0, cz = 0;
        if (use_com) {                                                   // K :86-111: every lane sums its own molecule, in slot order
            const int j = (int)((m >> 17) & 63u), n1 = (int)((m >> 23) & 63u);
            const int first = lane - j;
            mixed px = 0, py = 0, pz = 0, pm = 0;
            for (int k = 0; k < (bd.y & 255); k++) {
                if (k <= n1) {
                    const mixed um = im[first + k];
                    px += ix[first + k] * um; py += iy[first + k] * um; pz += iz[first + k] * um; pm += um;
                }
            }
            const mixed wq = rcp_(pm);
            cx = px * wq; cy = py * wq; cz = pz * wq;
            if (ke && j == 0 && lane < bd.n)                             // M v_com^2 (K :154)
                e->ke_com += ((double)cx * cx + (double)cy * cy + (double)cz * cz) * (double)pm;
        }
        t.cx = cx; t.cy = cy; t.cz = cz;
        if (ke) {                                                        // bins, as wke_kernel (K :138-200)
            const uint32_t role = m & 3u, g = (m >> 2) & 255u;
            const double rx = v.x - cx, ry = v.y - cy, rz = v.z - cz;
            double val = v.w != 0 ? (rx * rx + ry * ry + rz * rz) * (double)t.mass : 0.0;
            if (role == ROLE_DRUDE) {
                const int pl = lane + (int)((m >> 10) & 127u) - 64;
                const double dx = ix[pl] - v.x, dy = iy[pl] - v.y, dz = iz[pl] - v.z;
                const double mass1 = t.mass, mass2 = im[pl];
                const double mu = mass1 * mass2 * rcp_(mass1 + mass2);
                const double d = (dx * dx + dy * dy + dz * dz) * mu;
                e->ke_drude += d;
                val -= d;
            }
#pragma unroll
            for (int b = 0; b < GB; b++) e->ke_g[b] += (g == (uint32_t)b) ? val : 0.0;
        }
    }
    // second half (pass 2): rescale, half kick, drift, hard wall, stores.  The image holds the tile's kicked velocities and masses.
    __device__ __forceinline__ void finish(WStepIn<PREC>& t, const WaveBounds& bd) {
        mixed4 v = t.v;
        const uint32_t m = t.meta;
        const uint32_t role = m & 3u, g = (m >> 2) & 255u;
        const int pl = lane + (int)((m >> 10) & 127u) - 64;
        const mixed mass = t.mass, cx = t.cx, cy = t.cy, cz = t.cz;
        const mixed s_com = (mixed)s_scale[G], s_drude = (mixed)s_scale[G + 1], s_g = (mixed)s_scale[g];
        mixed px = t.p.x, py = t.p.y, pz = t.p.z;
        const real pq = t.p.w;
        if (PREC == TGNH_PREC_MIXED) { px += (mixed)t.c.x; py += (mixed)t.c.y; pz += (mixed)t.c.z; }
        // ---- A6: rescale (K :249-301 ; Ref :516-541), tile_body's expressions
        if (role == ROLE_NORMAL) {
            if (v.w != 0) {
                const mixed rx = v.x - cx, ry = v.y - cy, rz = v.z - cz;
                v.x = s_g * rx + s_com * (v.x - rx);
                v.y = s_g * ry + s_com * (v.y - ry);
                v.z = s_g * rz + s_com * (v.z - rz);
            }
        } else {
            const mixed ux = ix[pl], uy = iy[pl], uz = iz[pl], um = im[pl];      // partner velocity and mass
            const mixed rsx = v.x - cx, rsy = v.y - cy, rsz = v.z - cz;
            const mixed rpx = ux - cx, rpy = uy - cy, rpz = uz - cz;
            const mixed invTot = rcp_(mass + um);
            const mixed msf = invTot * mass, mpf = invTot * um;
            const mixed sdp = s_drude * mpf;
            v.x = s_g * (rsx * msf + rpx * mpf) + sdp * (rsx - rpx) + s_com * (v.x - rsx);
            v.y = s_g * (rsy * msf + rpy * mpf) + sdp * (rsy - rpy) + s_com * (v.y - rsy);
            v.z = s_g * (rsz * msf + rpz * mpf) + sdp * (rsz - rpz) + s_com * (v.z - rsz);
        }
        // ---- A7: half kick (K :307-365) and A8: drift (Ref :253-258 ; K :322-324, :450-452)
        if (v.w != 0) {
            const mixed c = fscale * v.w;
            v.x += c * force_as(t.fx, (mixed)0);
            v.y += c * force_as(t.fy, (mixed)0);
            v.z += c * force_as(t.fz, (mixed)0);
            px += dt * v.x; py += dt * v.y; pz += dt * v.z;
        }
        // ---- A10: hard wall (K :471-574 ; Ref :298-363), tile_body's arithmetic from the lane's own point of view
        if (hardwall) {
            wfence();                                                    // every lane has read its partner's old velocity
            ix[lane] = v.x; iy[lane] = v.y; iz[lane] = v.z;
            jx[lane] = px; jy[lane] = py; jz[lane] = pz;
            wfence();
            if (role != ROLE_NORMAL) {
                const mixed maxd = (mixed)a.max_dist, hws = (mixed)a.hw_scale;
                const mixed sxd = px - jx[pl], syd = py - jy[pl], szd = pz - jz[pl];     // self - partner
                const mixed d2 = sxd * sxd + syd * syd + szd * szd;
                if (d2 > maxd * maxd) {
                    const mixed4 uv = mk4(ix[pl], iy[pl], iz[pl], im[pl]);
                    const bool is_d = role == ROLE_DRUDE;
                    const mixed4 vel1 = is_d ? v : uv, vel2 = is_d ? uv : v;
                    const mixed dx = is_d ? sxd : -sxd, dy = is_d ? syd : -syd, dz = is_d ? szd : -szd;   // Drude - parent (K :487)
                    const mixed r = sqrt_(d2);
                    const mixed rInv = rcp_(r);
                    if (rInv * maxd < (mixed)0.5) atomicOr(a.status, 1u);     // Ref :311-312
                    const mixed bx = dx * rInv, by = dy * rInv, bz = dz * rInv;
                    const mixed mass1 = is_d ? mass : uv.w, mass2 = is_d ? uv.w : mass;
                    const mixed deltaR = r - maxd;
                    mixed deltaT = dt;
                    mixed dotvr1 = vel1.x * bx + vel1.y * by + vel1.z * bz;
                    const mixed vp1x = vel1.x - bx * dotvr1, vp1y = vel1.y - by * dotvr1, vp1z = vel1.z - bz * dotvr1;
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
                        px += bx * dr1; py += by * dr1; pz += bz * dr1;
                        v.x = vp1x + bx * dotvr1; v.y = vp1y + by * dotvr1; v.z = vp1z + bz * dotvr1;
                    } else {
                        px += bx * dr2; py += by * dr2; pz += bz * dr2;
                        v.x = vp2x + bx * dotvr2; v.y = vp2y + by * dotvr2; v.z = vp2z + bz * dotvr2;
                    }
                }
            }
        }
        wfence();                                                        // the next tile's image comes after this tile's reads
        if (lane < bd.n) {
            const int idx = bd.ws + lane;
            velm[idx] = v;
            if (PREC == TGNH_PREC_MIXED) {                               // K :457-458
                const float hx = (float)px, hy = (float)py, hz = (float)pz;
                posq[idx] = mk4((real)hx, (real)hy, (real)hz, pq);
                pcorr[idx] = make_float4((float)(px - hx), (float)(py - hy), (float)(pz - hz), 0.0f);
            } else {
                posq[idx] = mk4((real)px, (real)py, (real)pz, pq);
            }
        }
    }
};

}  // namespace tgnh
#endif
