/*
 * drude_tgnh.h -- C ABI of the MI355X (gfx950) DrudeTGNHIntegrator step kernel.
 *
 * This is the drop-in boundary: what an OpenMM-HIP `KernelImpl` for
 * `IntegrateDrudeTGNHStepKernel` (reference: openmmapi/include/openmm/DrudeTGNHKernels.h:48-74)
 * binds to.  Plain C types only; the caller owns the particle buffers (OpenMM's
 * posq / velm / force / posDelta device arrays), the library owns topology,
 * scratch and thermostat state.  All work is enqueued on the caller's HIP
 * stream; functions named *_get_* that return host values synchronise that
 * stream, nothing else does.  One handle per device; handles are not
 * thread-safe.  Every function returns a tgnh_status; tgnh_last_error() has
 * the message (the OpenMM glue turns it into an OpenMMException).
 *
 * Reference interfaces replaced (file:line in scychon/openmm_drudeNose):
 *   tgnh_create            IntegrateDrudeTGNHStepKernel::initialize
 *                          platforms/cuda/src/CudaDrudeTGNHKernels.cpp:75-282,
 *                          platforms/reference/src/ReferenceDrudeTGNHKernels.cpp:104-219
 *   tgnh_step_begin/_end   IntegrateDrudeTGNHStepKernel::execute
 *                          CudaDrudeTGNHKernels.cpp:284-408 split at the force call-out (:380);
 *                          ReferenceDrudeTGNHKernels.cpp:221-415 split at :384
 *   tgnh_get_kinetic_energy  ::computeKineticEnergy  CudaDrudeTGNHKernels.cpp:654-661,
 *                          ReferenceDrudeTGNHKernels.cpp:586-588
 *   tgnh_set_*             values the reference re-reads from the integrator every
 *                          step (CudaDrudeTGNHKernels.cpp:292,298,437-439,469)
 *   tgnh_destroy           ~CudaIntegrateDrudeTGNHStepKernel  CudaDrudeTGNHKernels.cpp:52-73
 */
#ifndef DRUDE_TGNH_H_
#define DRUDE_TGNH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGNH_ABI_VERSION 1

typedef int tgnh_status;
enum {
    TGNH_OK = 0,
    TGNH_ERR_ARG = -1,            /* bad argument / size */
    TGNH_ERR_GROUP_MISMATCH = -2, /* CudaDrudeTGNHKernels.cpp:145-146, :192-193 */
    TGNH_ERR_HARDWALL = -3,       /* ReferenceDrudeTGNHKernels.cpp:311-312 */
    TGNH_ERR_UNSUPPORTED = -4,    /* what the reference itself cannot run (a massless pair member, a molecule without mass under the COM group,
                                     dualNH without a pair), sizes beyond an index's or a bin's reach, a flag the topology cannot take;
                                     NOT a topology the tiles cannot hold: that steps on the gather path (tgnh_get_step_path) */
    TGNH_ERR_HIP = -5,            /* a HIP runtime call failed */
    TGNH_ERR_STATE = -6           /* call order (buffers not bound, ...) */
};

/* Semantic mode: the two platform implementations differ. */
enum { TGNH_MODE_DUALNH = 0,      /* platforms/reference semantics */
       TGNH_MODE_TGNH = 1 };      /* platforms/cuda semantics (temperature groups + COM) */

/* OpenMM precision modes: real = posq type, mixed = velm/posDelta type. */
enum { TGNH_PREC_SINGLE = 0,      /* float4 posq, float4 velm */
       TGNH_PREC_MIXED = 1,       /* float4 posq + float4 posqCorrection, double4 velm */
       TGNH_PREC_DOUBLE = 2 };    /* double4 posq, double4 velm */

/* flags for tgnh_desc.flags */
enum { TGNH_FLAG_DEFER_SCALE = 2 };    /* the end-of-step rescale AND the second half kick are left pending and folded
                                        * into the next step's first pass: between tgnh_step_end and the next
                                        * tgnh_step_begin velm lags and the force buffer must stay as it is; tgnh_flush
                                        * makes velm the reference's end-of-step state.  (1 is reserved.) */
enum { TGNH_FLAG_WAVE_TILES = 8 };     /* run the kinetic-energy passes and the one-launch deferred step over wave tiles (one
                                        * wavefront = <= 64 slots) whenever the topology has them -- every molecule and
                                        * pair inside a wavefront, <= 8 temperature groups -- whatever their fill.  Without the
                                        * flag the library takes them only where they are >= 90 % full (60-slot water tiles: yes;
                                        * 35-slot cations: no, the 512-slot tile kernels are faster there).  Same integrator
                                        * either way. */
enum { TGNH_FLAG_RESIDENT_STEP = 4 };  /* whole thermostat halves in ONE launch whose work-groups meet on the device
                                        * (step_kernel).  Alone: the reference's pass structure, two launches per
                                        * step (KE + chain + rescale, kick, drift | kick + KE + chain + rescale), velocities
                                        * never lag -- usable wherever the plain structure is.  With DEFER_SCALE:
                                        * tgnh_step_end launches nothing and the next tgnh_step_begin runs that end half and
                                        * its own begin half in one launch.  Either way the handle needs the device to itself
                                        * while stepping (see tgnh_set_resident_share); <= 8 temperature groups, no
                                        * collective hook, and chains of 1-4 links with DEFER_SCALE on a topology with wave tiles
                                        * (wstep_kernel; not DUALNH's coupled chain of 2-4 links, useDrudeNHChains = 0), one-link
                                        * chains otherwise (step_kernel) -- else it quietly steps with
                                        * the tile launches (tgnh_get_resident_kernel says which).
                                        * With a mailbox exchange attached and DEFER_SCALE, state queries between steps are
                                        * collective over the ranks. */

enum { TGNH_FLAG_TRUST_STATE_CHANGED = 16 };  /* The reference's pass structure (velocities never lag, no TGNH_FLAG_DEFER_SCALE)
                                        * without the first half step's kinetic-energy pass: after the end half's rescale every bin
                                        * is exactly s^2 times the bin its chain started from, and the chain has tracked that product
                                        * (CudaDrudeTGNHKernels.cpp:574), so the next tgnh_step_begin / _begin_kick starts its chain
                                        * from it -- one launch instead of KE + chain + rescale.  The caller vouches that NOTHING
                                        * writes velm between tgnh_step_end / _end_thermo and the next begin without
                                        * tgnh_state_changed (DrudeTGNHIntegrator.cpp:166-170): setVelocities does call it, OpenMM's
                                        * CMMotionRemover and AndersenThermostat edit velocities in updateContextState without -- the
                                        * glue sets the flag only for a System that holds neither.  Invalidated (the next half
                                        * recomputes) by tgnh_state_changed, tgnh_bind_buffers, tgnh_set_thermostat_state,
                                        * tgnh_set_global_dof_terms, the split entry points that write velocities, attaching an
                                        * exchange; ignored when sharded (hook, RCCL, mailboxes), with DEFER_SCALE, and for a topology
                                        * in which a molecule spans two temperature groups (tgnh_get_pending_state bit 9 shows
                                        * whether the next half will carry over).  A hipGraph of steps recorded while carrying over
                                        * holds no KE pass: replay it only while the promise holds. */

enum { TGNH_FLAG_GATHER = 32 };        /* step on the GATHER path (tgnh_get_step_path) whatever the topology: the reference's own
                                        * un-fused kernels by global index, which the library otherwise takes only for what its
                                        * tiles cannot hold.  For comparisons and as a fallback; several times slower per slot. */

typedef struct tgnh_desc {
    uint32_t struct_size;         /* sizeof(tgnh_desc), ABI check */
    int32_t mode;                 /* TGNH_MODE_* */
    int32_t precision;            /* TGNH_PREC_* */
    int32_t flags;                /* TGNH_FLAG_* ; any other bit: TGNH_ERR_ARG (a binding built against a newer header) */
    int32_t device;               /* HIP device ordinal; -1 = host-only handle (topology, tiles, dof queries; no launches) */
    int32_t num_particles;        /* N: particle slots owned by this handle */
    int32_t padded_num_particles; /* stride of the 3 force planes (OpenMM PADDED_NUM_ATOMS), >= N and <= 715 827 882 (3 x stride in 32-bit indices, as K :318-320) */
    int32_t num_pairs;            /* P: Drude pairs, DrudeForce order */
    int32_t num_groups;           /* G: getNumTempGroups() (TGNH mode) */
    int32_t num_residues;         /* R: getNumResidues()   (TGNH mode) */
    int32_t num_constraints;
    int32_t has_cm_motion_remover;
    const double* mass;           /* [N] System::getParticleMass */
    const int32_t* pair_drude;    /* [P] DrudeForce::getParticleParameters p  */
    const int32_t* pair_parent;   /* [P] DrudeForce::getParticleParameters p1 */
    const int32_t* group;         /* [N] getParticleTempGroup (TGNH mode; ignored in DUALNH -- the Reference platform has no temperature groups -- and may be NULL there) */
    const int32_t* resid;         /* [N] getParticleResId     (TGNH mode; may be NULL in DUALNH) */
    const int32_t* constraint_i;  /* [num_constraints] or NULL */
    const int32_t* constraint_j;
    double kB;                    /* OpenMM BOLTZ, kJ/mol/K */
    double temperature, coupling_time;
    double drude_temperature, drude_coupling_time;
    double step_size;
    int32_t drude_steps_per_real_step;
    int32_t num_nh_chains;
    int32_t use_drude_nh_chains;
    int32_t use_com_temp_group;
    double max_drude_distance;    /* 0 = hard wall off */
} tgnh_desc;

typedef struct tgnh_context* tgnh_handle;

/* What is reproducible bit for bit (no atomics anywhere in a sum; every order of additions is fixed by the topology, the grid
 * and the direction of the sweep):
 *   - the same handle configuration, the same state and the same step counter give the same bits from there on: the sweep
 *     direction, which orders the additions of the kinetic-energy sums, is a function of the step counter, and queries
 *     (tgnh_get_*, tgnh_compute_kinetic_energies, tgnh_flush) neither change it nor anything else of the trajectory -- so a
 *     run restored from a checkpoint (thermostat arrays + tgnh_set_time) continues bit for bit, and a query asked twice
 *     returns the same bits (the plain kinetic energy of tgnh_get_kinetic_energy included);
 *   - the ranks of one sharded run hold bit-identical thermostats whatever each of them is asked in between: every rank adds
 *     the same world x NT sums in rank order (mailboxes) or receives the same all-reduced values (RCCL) and runs the same
 *     chain arithmetic, in the standalone chain kernel or inside a streaming launch;
 *   - different pass structures (flags), grids (devices with another number of compute units) or shardings of the same
 *     system sum in different orders: they agree to rounding (1e-12 over 100 steps), not bit for bit. */

/* Collective hook for particle-sharded runs: in-place sum of `count` doubles
 * at device pointer `buf` over all ranks, enqueued on `stream` (an RCCL
 * ncclAllReduce in the glue / a torch.distributed all_reduce in the harness).
 * Must not synchronise the host. */
typedef int (*tgnh_allreduce_fn)(void* buf, int count, void* stream, void* user);

const char* tgnh_last_error(void);
int tgnh_abi_version(void);

tgnh_status tgnh_create(const tgnh_desc* desc, tgnh_handle* out);
tgnh_status tgnh_destroy(tgnh_handle h);

/* OpenMM device arrays (raw device pointers).  posq_correction only in MIXED
 * mode, pos_delta only for the constrained (split) path; pass NULL otherwise.
 *   posq   real4  [N]  (x,y,z,q)
 *   velm   mixed4 [N]  (vx,vy,vz,1/m)   -- w is read as the inverse mass
 *   force  int64  [3*padded]  fixed point x 2^32, planes x|y|z
 *   pos_delta mixed4 [N]
 * When a pointer differs from the one bound before, what the runtime knows of the allocation behind each pointer is checked
 * against what the launches touch (TGNH_ERR_ARG: a float4 array bound as double4, N for padded).  A call with the five pointers
 * already bound returns at once WITHOUT that check (the glue binds at every step): an array freed and allocated again, smaller,
 * at the same address is the caller's to rebind through a NULL-free change of any pointer, or to not do. */
tgnh_status tgnh_bind_buffers(tgnh_handle h, void* posq, void* posq_correction, void* velm,
                              const void* force, void* pos_delta);

/* Integrator scalars re-read every step by the reference. */
tgnh_status tgnh_set_step_size(tgnh_handle h, double dt);
tgnh_status tgnh_set_drude_steps_per_real_step(tgnh_handle h, int n);
tgnh_status tgnh_set_max_drude_distance(tgnh_handle h, double d);

/* Particle sharding: dof terms are additive over ranks.  terms = per-thermostat
 * degrees of freedom before the global CMMotionRemover correction
 * (NT values, see tgnh_get_num_thermostats).  Sum them over ranks and hand them back. */
tgnh_status tgnh_get_local_dof_terms(tgnh_handle h, double* terms, int* count);
tgnh_status tgnh_set_global_dof_terms(tgnh_handle h, const double* terms, int count);
tgnh_status tgnh_set_allreduce(tgnh_handle h, tgnh_allreduce_fn fn, void* user);

/* The collective done by the library itself: one ncclAllReduce(sum, ncclDouble, NT values, in place) per thermostat half
 * step -- per time step with TGNH_FLAG_DEFER_SCALE -- enqueued on the step's stream between the row sum and the chain
 * (the reference integrates on one device only, platforms/cuda/src/CudaDrudeTGNHKernelFactory.cpp:62: this is the exchange
 * north_star adds, "a single RCCL all-reduce over xGMI of the per-group KE scalars each step").  Either hand over a
 * communicator the caller already has (an ncclComm_t of the RCCL this library is linked against), or let the library make
 * its own: rank 0 calls tgnh_rccl_unique_id and passes the TGNH_RCCL_ID_BYTES to every rank by whatever means the caller
 * has (MPI, a file, torch.distributed), then EVERY rank calls tgnh_rccl_init (collective; the handle's device must be
 * current-able).  Replaces a tgnh_set_allreduce hook; tgnh_rccl_shutdown (or tgnh_destroy) destroys a communicator the
 * library made.  Capturable into a hipGraph like every launch of a step. */
#define TGNH_RCCL_ID_BYTES 128
tgnh_status tgnh_rccl_unique_id(void* id_out);
tgnh_status tgnh_rccl_init(tgnh_handle h, int world, int rank, const void* id);
tgnh_status tgnh_set_rccl_comm(tgnh_handle h, void* nccl_comm);
tgnh_status tgnh_rccl_shutdown(tgnh_handle h);
/* TGNH_FLAG_RESIDENT_STEP: this handle may fill 1/share of the device's resident work-group slots (default 1 = all of
 * them).  Several handles that step concurrently on one device (replicas, or the ranks of a rehearsal on one GPU) must
 * share it, or their launches wait for each other's work-groups until the meeting times out (status bit 3). */
tgnh_status tgnh_set_resident_share(tgnh_handle h, int share);
/* Work-groups of the resident step kernel per compute unit that tgnh_create found resident together (a census launch checks
 * the occupancy query); 0 = the handle steps the DEFER_SCALE way (flag not set, or nothing passed the census). */
tgnh_status tgnh_get_resident_work_groups(tgnh_handle h, int* per_compute_unit);
/* How this handle steps: *gather = 0 the tiled kernels (every Drude partner and molecular centre of mass an on-chip look-up inside
 * a tile of <= 512 consecutive slots, <= 32 temperature groups), 1 the GATHER path -- the reference's own un-fused kernels by
 * global index (drudeTGNH.cu:82-301, 307-365, 435-574: normalParticles / pairParticles / particleResId lists, bins sized G + 2),
 * taken for what the tiles cannot hold: a Drude particle more than a tile from its parent, pairs overlapping so densely in a long
 * molecule that no cut between two of them lies within a tile's reach, more than 32 temperature groups (up to 2046), chains
 * too long for the on-chip forms; 2 the same with its own row sum and chain kernels (more than 34 thermostats).  Much slower per
 * slot (every look-up a global load) and always in the reference's pass structure: TGNH_FLAG_DEFER_SCALE, _RESIDENT_STEP,
 * _TRUST_STATE_CHANGED are ignored on it.  *reason (may be NULL): why, "" for the tiled path; valid until tgnh_destroy.
 * SHARDED RUNS: the ranks exchange once per thermostat half step they run, and a rank on the gather path runs the reference's
 * two halves per step whatever its flags say -- so all ranks must be on the same path: if tgnh_get_step_path reports the gather
 * path on ANY rank (agree on it with the collective at hand), create every rank's handle with TGNH_FLAG_GATHER.  Ranks that
 * disagree wait for exchanges their peers never send: the mailbox exchange times out (status bit 2), a collective hangs. */
tgnh_status tgnh_get_step_path(tgnh_handle h, int* gather, const char** reason);
/* Which kernel this handle's next tgnh_step_begin runs as its one launch: 0 none (the streaming launches), 1 step_kernel (512-slot
 * tiles, one-link chains), 2 wstep_kernel (a whole deferred step over wave tiles, chains of 1-4 links). */
tgnh_status tgnh_get_resident_kernel(tgnh_handle h, int* which);

/* Mailbox exchange: the same all-reduce of the NT kinetic-energy sums, done by the integrator's own kernels with
 * plain stores into every peer's mailbox over xGMI (no collective launch on the step's critical path).  Optional;
 * when attached it replaces the hook above.  One rank per GPU (or several handles in one process):
 *   create  allocates this rank's mailbox and returns its IPC handle (TGNH_XCHG_HANDLE_BYTES, for peers in other
 *           processes) and its device pointer (for peers in this process);
 *   attach  takes every rank's IPC handle, rank-major (own entry ignored) -- or every rank's pointer;
 * then all ranks step in lockstep.  Sums are added in rank order on every rank: bit-identical thermostats.
 * A peer that never answers sets status bit 2 after a bounded wait (tgnh_get_status_flags); nothing hangs.
 * Teardown across processes: every rank detaches (unmaps the peers' mailboxes), then -- after a barrier of the
 * caller's -- destroys its handle (frees its own mailbox). */
#define TGNH_XCHG_HANDLE_BYTES 64
tgnh_status tgnh_exchange_create(tgnh_handle h, int world, int rank, void* ipc_handle_out, void** mailbox_out);
tgnh_status tgnh_exchange_attach(tgnh_handle h, const void* ipc_handles);
tgnh_status tgnh_exchange_attach_pointers(tgnh_handle h, void* const* mailboxes);
tgnh_status tgnh_exchange_detach(tgnh_handle h);
/* How long the waits of the mailbox exchange took on this rank since the last call (or attach): per exchange, the time from
 * "this rank's sums are complete" to "every rank's sums are here" as seen by work-group 0 of the waiting launch, measured on
 * the device (wall_clock64, 10 ns ticks) -- the figure that says WHERE a sharded run loses time when it scales worse than the
 * one-GPU ceiling: a rank that waits long is waiting for a slower peer or for the link.  Synchronises `stream`; resets. */
tgnh_status tgnh_exchange_wait_stats(tgnh_handle h, void* stream, double* mean_us, double* max_us, int64_t* exchanges);

/* One time step, split at the force call-out:
 *   begin: [KE -> chain ->] rescale, half kick, drift, hard wall
 *   (caller: virtual sites, calcForcesAndEnergy)
 *   end:   half kick, KE -> chain -> rescale, time += dt */
tgnh_status tgnh_step_begin(tgnh_handle h, void* stream);
tgnh_status tgnh_step_end(tgnh_handle h, void* stream);
/* Split variants around OpenMM's constraint call-outs (posDelta path):
 *   begin_kick:  [KE -> chain ->] rescale, half kick, posDelta = dt*v        (CudaDrudeTGNHKernels.cpp:336-360)
 *   (caller: applyConstraints on posDelta)
 *   begin_move:  pos += posDelta, v = posDelta/dt, hard wall                 (:366-376)
 *   end_kick:    half kick                                                   (:384-388)
 *   (caller: applyVelocityConstraints)
 *   end_thermo:  KE -> chain -> rescale, time += dt                          (:394-406) */
tgnh_status tgnh_step_begin_kick(tgnh_handle h, void* stream);
tgnh_status tgnh_step_begin_move(tgnh_handle h, void* stream);
tgnh_status tgnh_step_end_kick(tgnh_handle h, void* stream);
tgnh_status tgnh_step_end_thermo(tgnh_handle h, void* stream);
/* Apply the half kick and rescale still pending (TGNH_FLAG_DEFER_SCALE) so velm is the
 * reference's end-of-step state; call before anything else reads velm. */
tgnh_status tgnh_flush(tgnh_handle h, void* stream);
/* A caller that captured `nsteps` steps into a hipGraph and replays it tells the handle here: the host-side
 * clock and step count advance only when the step functions are called, not when a graph is replayed.
 * Determinism: the sweep direction of a step's launches -- which orders the additions of its kinetic-energy sums -- is fixed from
 * the step counter when the step is ENQUEUED (step k starts in direction k & 1, see "What is reproducible bit for bit" above), so a recording bakes
 * its directions in.  A graph of an EVEN number of steps replays the eager loop's launches exactly; a graph of an odd number
 * does so on every other replay only -- record two graphs back to back and replay them in turn (HipContext.capture_steps does)
 * if the trajectory is to be the eager one bit for bit.  Either way the result is correct to rounding. */
tgnh_status tgnh_note_replayed_steps(tgnh_handle h, int nsteps);
/* What the handle still owes the trajectory, as a bit set (inspection only; a caller that records steps into a hipGraph
 * checks that the set is the same before and after the recorded steps -- a graph replays launches, not decisions):
 * bit 0 the end half of the last step waits for the next tgnh_step_begin (RESIDENT_STEP), 1 velm lags by the scale factors,
 * 2 velm lags by the second half kick, 3 the next step's first thermostat half has already run (DEFER_SCALE), 4 the summed
 * kinetic energies wait for the next rescale launch to run the chain, 5 the partial rows are not summed yet, 6 an exchange is
 * sent but not yet waited for, 7 the advanced thermostat block sits in the staging copy, 8 direction of the next sweep,
 * 9 the next thermostat half step starts from the kinetic energies the last one left (TGNH_FLAG_TRUST_STATE_CHANGED). */
tgnh_status tgnh_get_pending_state(tgnh_handle h, uint32_t* bits);
/* Restore the clock of a checkpointed run (time, stepCount: ReferenceDrudeTGNHKernels.cpp:413-414, CudaDrudeTGNHKernels.cpp:405-406). */
tgnh_status tgnh_set_time(tgnh_handle h, double time, int64_t step_count);
/* Velocities were changed behind the integrator's back (setVelocities, CMMotionRemover,
 * barostat): cached kinetic energies are stale.  DrudeTGNHIntegrator.cpp:166-170 */
tgnh_status tgnh_state_changed(tgnh_handle h);

/* Host-visible results (synchronise `stream`).
 * tgnh_get_kinetic_energy: TGNH mode = the cached 1/2 sum of the last thermostat half step's bins when ke_sum_valid,
 * else 1/2 sum m v^2 (CudaDrudeTGNHKernels.cpp:654-658).  DUALNH mode = 1/2 sum m (v + F dt/2m)^2, the Reference
 * platform's half-step-shifted energy (ReferenceDrudeTGNHKernels.cpp:70-98) WITHOUT its constraint projection, which is
 * a call-out to the host's constraint solver: with constraints present project the shifted velocities first
 * (tests/test_reference_water_gpu.py::reference_platform_kinetic_energy shows the sequence). */
tgnh_status tgnh_get_kinetic_energy(tgnh_handle h, int ke_sum_valid, void* stream, double* out);
tgnh_status tgnh_get_num_thermostats(tgnh_handle h, int* count);               /* NT: TGNH G+2 = [groups.., COM, Drude]; DUALNH 3 = [real, unused, Drude] */
tgnh_status tgnh_get_last_kinetic_energies(tgnh_handle h, void* stream, double* ke);   /* no 1/2; before the chain */
tgnh_status tgnh_get_last_scale_factors(tgnh_handle h, void* stream, double* scale);
/* bit0: a Drude beyond 2x the hard wall; bit1: the harness SHAKE did not converge; bit2: a mailbox exchange timed out;
 * bit3: the work-groups of a resident step did not all meet; bit4: a kinetic-energy pass that sums its own rows (tail sum) did not
 * receive every row -- the sums were left as NaN -- or a chain was handed a NaN sum (with an all-reduce attached the NaN of one
 * rank's failed pass reaches every rank, and every rank sets the bit itself).  *flags is always filled in.  bit2, bit3, bit4 -- and bit0 in DUALNH mode, where the Reference platform throws
 * (ReferenceDrudeTGNHKernels.cpp:311-312) -- are FAILURES and sticky: once the host has seen one (here, at any other
 * tgnh_get_*, at tgnh_exchange_detach, or through the read-back the library enqueues behind every 64th step) every later
 * tgnh_step_*, tgnh_flush and tgnh_get_* returns TGNH_ERR_STATE / TGNH_ERR_HARDWALL with the message in tgnh_last_error(). */
tgnh_status tgnh_get_status_flags(tgnh_handle h, void* stream, uint32_t* flags);
tgnh_status tgnh_get_time(tgnh_handle h, double* time, int64_t* step_count);
tgnh_status tgnh_get_dof(tgnh_handle h, double* dof, double* nkt);

/* Thermostat state (checkpoint/resume, absent in the reference: SURVEY.md 5).
 * which: 0 eta, 1 etaDot, 2 etaDotDot, 3 etaMass.  Layout: DUALNH = the
 * reference's interleaved vectors; TGNH = [thermostat][link], etaDot rows of C+1. */
tgnh_status tgnh_get_thermostat_len(tgnh_handle h, int which, int* len);
tgnh_status tgnh_get_thermostat_state(tgnh_handle h, int which, void* stream, double* out);
tgnh_status tgnh_set_thermostat_state(tgnh_handle h, int which, void* stream, const double* in);

/* Topology the kernels were built from, for bit-exact index parity (A1).
 * which: 0 normalParticles, 1 pair drude, 2 pair parent, 3 particleTempGroup,
 * 4 particleResId, 5 particlesInResidues.count, 6 particlesInResidues.first,
 * 7 tile starts (num_tiles+1), 8 packed per-slot meta words, 9 wave tiles as (first slot, largest molecule) pairs
 * (one more than tiles; none: a molecule or pair longer than a wavefront), 10 the wave tiles' per-slot words,
 * 11 per 512-slot tile: period | molecules per period << 8 | pattern << 16, 12 per wave tile: period | pattern << 8, where
 * the tile repeats one kind of molecule (or a few) and its kernels form the per-slot words from a pattern instead of
 * reading them (0: they read them), 13 / 14 the patterns of the two, 64 words each. */
tgnh_status tgnh_get_topology_len(tgnh_handle h, int which, int* len);
tgnh_status tgnh_get_topology(tgnh_handle h, int which, int32_t* out);
/* The sizes the launches of this handle are bound by, against the sizes of what the library allocated for them (inspection;
 * also answered by a host-only handle, for the largest grid any device could give it).  The library checks the same figures
 * before every launch (TGNH_ERR_STATE "internal: ..." instead of a launch that would run off a buffer).  out[8]:
 *   0 512-slot tiles, 1 wave tiles, 2 entries of the wave-tile table (wave tiles + 1; 0: none), 3 the largest grid a streaming
 *   launch of this handle takes, 4 partial rows allocated for such launches, 5 words allocated for tagged rows (0: none),
 *   6 words a launch of grid [3] with this handle's thermostats may write there (0: it never does), 7 thermostats NT. */
tgnh_status tgnh_get_launch_bounds(tgnh_handle h, int32_t out[8]);

/* Pieces of the step, exposed for parity tests of the single kernels. */
tgnh_status tgnh_compute_kinetic_energies(tgnh_handle h, void* stream);          /* A3/A4 only -> last_kinetic_energies */
tgnh_status tgnh_half_kick(tgnh_handle h, void* stream);                         /* A7 only */

/* Harness force (bench/test workload, not part of the reference): Drude spring
 * + tether to sites x0 (real4 [N] device array: x,y,z and w = 1 for a tethered slot, 0 otherwise),
 * written in OpenMM's fixed-point layout into `force_out` (int64 [3*padded]). */
tgnh_status tgnh_harness_force(tgnh_handle h, const void* x0, double k_drude, double k_tether,
                               void* force_out, void* stream);
/* Packs the sites once (synchronous; x0 as above, its writes complete): the tethered slots' sites only, 12 or 24 B each,
 * and one byte per slot in place of the meta word.  Afterwards tgnh_harness_force / tgnh_run_harness* accept x0 = NULL
 * and read the packed form (the same forces bit for bit). */
tgnh_status tgnh_harness_pack_sites(tgnh_handle h, const void* x0);
/* A hint for the next tgnh_harness_pack_sites: the box repeats one molecule of `mol_slots` slots (<= 64) on a simple cubic lattice --
 * this handle's molecule j is molecule m = first_molecule + j of the box (a shard starts in the middle of it), at
 * (m / side^2, (m / side) mod side, m mod side) x spacing, slot k of it at that point + geom[k] (doubles [mol_slots][3]),
 * rounded once to the position type -- as the synthetic water boxes are built.  pack_sites CHECKS every tethered site against that
 * formula, bit for bit, and every slot's role / partner / tether flag against molecule 0's; if all hold, the force kernel forms
 * sites and flags from the slot index and reads nothing but positions (56 B per slot in mixed precision: what it must read and
 * write anyway; 65 B with packed sites); if not, the hint is dropped and the sites are packed as before.  mol_slots = 0 clears it. */
tgnh_status tgnh_harness_lattice_hint(tgnh_handle h, int mol_slots, int side, double spacing, const double* geom, int first_molecule);
/* 0 explicit x0 / nothing packed yet, 1 packed sites, 2 lattice sites (what the last tgnh_harness_pack_sites ended in). */
tgnh_status tgnh_harness_sites_kind(tgnh_handle h, int* kind);
/* nsteps x { step_begin, harness force into the bound force buffer, step_end }
 * enqueued back to back with no host synchronisation. */
/* The step loop with NO call-out: nsteps x (tgnh_step_begin, tgnh_step_end) on whatever the bound force buffer holds.  For
 * timing the integrator's own launches between two synchronisations (bench.py's integrator_only leg zeroes the buffer first:
 * a frozen spring force would drive every Drude particle through its hard wall within ten steps).  Not an integrator of anything. */
tgnh_status tgnh_run_steps(tgnh_handle h, int nsteps, void* stream);
tgnh_status tgnh_run_harness(tgnh_handle h, const void* x0, double k_drude, double k_tether,
                             int nsteps, void* stream);

/* Harness call-outs of the constrained (split) path -- stand-ins for OpenMM's applyConstraints /
 * applyVelocityConstraints / computeVirtualSites (CudaDrudeTGNHKernels.cpp:363, :391, :377), not part of the reference.
 * Clusters: [n][4] slot indices (-1 = unused) and [n][6] distances for the pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
 * of a cluster, 0 = unconstrained.  Virtual sites: [n][4] = (site, p1, p2, p3), [n][3] weights (three-particle average). */
tgnh_status tgnh_harness_set_clusters(tgnh_handle h, int n, const int32_t* atoms, const double* dist);
tgnh_status tgnh_harness_shake_positions(tgnh_handle h, double tol, void* stream);    /* SHAKE on posDelta */
tgnh_status tgnh_harness_shake_velocities(tgnh_handle h, double tol, void* stream);   /* velocity stage on velm */
tgnh_status tgnh_harness_set_virtual_sites(tgnh_handle h, int n, const int32_t* atoms, const double* weights);
tgnh_status tgnh_harness_virtual_sites(tgnh_handle h, void* stream);
/* nsteps x { begin_kick, SHAKE, begin_move, virtual sites, harness force, end_kick, [TGNH: velocity stage], end_thermo } */
tgnh_status tgnh_run_harness_constrained(tgnh_handle h, const void* x0, double k_drude, double k_tether,
                                         double tol, int nsteps, void* stream);

/* The call-outs of the reference's own testWater (platforms/reference/tests/TestReferenceDrudeTGNHIntegrator.cpp:111-166),
 * harness only: its force field -- NonbondedForce with reaction field, cutoff `cutoff` in a cubic box of edge `box`, +
 * DrudeForce + the M site's force spread over O, H1, H2, for slots laid out O, D, H1, H2, M per molecule -- written into
 * `force_out` in OpenMM's fixed-point layout, and OpenMM's CMMotionRemover on the bound velocities. */
tgnh_status tgnh_harness_water_force(tgnh_handle h, double box, double cutoff, void* force_out, void* stream);
tgnh_status tgnh_harness_remove_cm_motion(tgnh_handle h, void* stream);

/* Per-kernel launch statistics gathered with HIP events on `stream` while
 * enabled (bench.py's live roofline).  kernel: 0 scale+kick+drift, 1 kick+KE,
 * 2 rescale, 3 KE, 4 chain, 5 harness force, 6 other, 7 resident step.  on = 1: every kernel; on = 2 + k: kernel k only
 * (two event records per step instead of eight, for timing inside a throughput measurement); 0: off. */
tgnh_status tgnh_timing_enable(tgnh_handle h, int on);
tgnh_status tgnh_timing_read(tgnh_handle h, int kernel, double* total_ms, int64_t* launches);
/* Algorithmic HBM bytes of one launch of `kernel` (SURVEY.md 8d model: state arrays only). */
tgnh_status tgnh_algorithmic_bytes(tgnh_handle h, int kernel, double* bytes);

#ifdef __cplusplus
}
#endif
#endif
