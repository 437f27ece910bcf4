// Host-visible launch interface of vsmpc_kernels.hip (internal to the library; the public
// boundary is include/vsmpc.h).
#pragma once
#include <hip/hip_runtime.h>

#include "vsmpc_device.hpp"

namespace vsmpc {

enum Variant { VARIANT_NONE = 0, VARIANT_PAPER = 1, VARIANT_H2X = 2 };

int select_variant(int n_iter, int n_iter_small, int control_horizon);
const char* variant_kernel_name(int variant);
int variant_condensed_dim(int variant);
size_t variant_workspace_doubles(int variant);  // per-instance global workspace of the factor (0 = lives in LDS)

hipError_t launch_solve(int variant, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                        int* d_status, int* d_iters, double* dbgM, double* dbgL, unsigned long long* stamps,
                        double* ws, hipStream_t stream);
hipError_t launch_linearize(int variant, const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj,
                            double* Bt, double* c, hipStream_t stream);

hipError_t launch_kinematics(const double* d_kin, int batch, double* d_out, hipStream_t stream);

// closed-loop rollout (vsmpc_rollout.hip)
struct RolloutDev {
    int n_in, n_ref, ratio, n_traj, n_alpha;
    double period_mpc, alpha_dt;
};
struct RolloutCtl {   // device-resident per-run control block of the rollout
    double* log;      // [ticks of this run][batch][VSMPC_ROLLOUT_LOG] or nullptr
    int tick_base;    // tick counter at the start of the run
    int pad;
};
hipError_t launch_record(const RolloutDev& rd, int batch, const double* state, const double* params, const int* tick,
                         const double* traj_pos, const double* traj_vel, const double* traj_alpha, double* rec,
                         hipStream_t stream);
hipError_t launch_advance(const RolloutDev& rd, int batch, double* state, const double* params, int* tick, const double* fm,
                          const int* status, const int* iters, const double* traj_alpha, const RolloutCtl* ctl, int substeps,
                          const double* traj_pos, const double* traj_vel, double* rec_next, hipStream_t stream);

}  // namespace vsmpc
