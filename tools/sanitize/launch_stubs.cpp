// Host-only sanitizer build: the kernel launchers of tgnh_kernels.hip / tgnh_harness.hip replaced by stubs that
// fail, so that tgnh_host.cpp (topology, tiles, dof, orchestration) can be compiled with g++ -fsanitize and
// exercised through host-only handles (device -1) on a machine without a GPU.  Nothing here ships.
#include "../../openmm_drudenose_amd/csrc/tgnh_internal.h"

namespace tgnh {
hipError_t launch_tile(int, int, int, const TileArgs&, int, size_t, hipStream_t) { return hipErrorNoDevice; }
int tile_blocks_per_cu(int, int, int, size_t, bool) { return 2; }
hipError_t launch_step(int, int, int, const TileArgs&, int, size_t, hipStream_t) { return hipErrorNoDevice; }
int step_blocks_per_cu(int, int, int, size_t) { return 2; }
int step_kind_ops2(int) { return 0; }
hipError_t launch_wstep(int, int, bool, const TileArgs&, int, hipStream_t) { return hipErrorNoDevice; }
int wstep_blocks_per_cu(int, int, bool) { return 2; }
hipError_t launch_wke(int, int, int, const TileArgs&, int, hipStream_t) { return hipErrorNoDevice; }
int wke_blocks_per_cu(int, int, int) { return 5; }
hipError_t launch_chain(const ChainArgs&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_big_com(int, const BigComArgs&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_force(int, const ForceArgs&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_plain_ke(int, const void*, const long long*, int, int, double, double*, hipStream_t) { return hipErrorNoDevice; }
size_t tile_lds_bytes(int, int, bool, bool) { return 0; }
hipError_t launch_gather_com(int, const GatherArgs&, hipStream_t) { return hipErrorNoDevice; }
int gather_ke_grid(const GatherArgs&) { return 1; }
hipError_t launch_gather_ke(int, const GatherArgs&, int, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_gather_rowsum(const double*, int, int, double*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_gather_chain(const ChainArgs&, double*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_gather_update(int, const GatherArgs&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_gather_force(int, const GatherArgs&, const void*, long long*, double, double, hipStream_t) { return hipErrorNoDevice; }
}  // namespace tgnh

extern "C" {
tgnh_status tgnh_harness_set_clusters(tgnh_handle, int, const int32_t*, const double*) { return TGNH_ERR_STATE; }
tgnh_status tgnh_harness_shake_positions(tgnh_handle, double, void*) { return TGNH_ERR_STATE; }
tgnh_status tgnh_harness_shake_velocities(tgnh_handle, double, void*) { return TGNH_ERR_STATE; }
tgnh_status tgnh_harness_set_virtual_sites(tgnh_handle, int, const int32_t*, const double*) { return TGNH_ERR_STATE; }
tgnh_status tgnh_harness_virtual_sites(tgnh_handle, void*) { return TGNH_ERR_STATE; }
tgnh_status tgnh_run_harness_constrained(tgnh_handle, const void*, double, double, double, int, void*) { return TGNH_ERR_STATE; }
}
