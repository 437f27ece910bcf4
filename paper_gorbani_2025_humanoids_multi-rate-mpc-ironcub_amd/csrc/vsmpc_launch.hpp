// Host-visible launch interface of vsmpc_kernels.hip (internal to the library; the public
// boundary is include/vsmpc.h).
#pragma once
#include <hip/hip_runtime.h>

#include "vsmpc_device.hpp"

struct vsmpc_jet;   // include/vsmpc_jet.h

namespace vsmpc {

// Entry points that must run on the handle's device switch to it for their own duration only: the caller's current device
// is put back on every return path (a library call that silently re-targets the caller's later HIP calls is a bug).
struct DeviceScope {
    int caller = -1, target = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device) : target(device) {
        err = hipGetDevice(&caller);
        if (err == hipSuccess && caller != target) err = hipSetDevice(target);
    }
    ~DeviceScope() { if (caller >= 0 && caller != target) (void)hipSetDevice(caller); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

enum Variant { VARIANT_NONE = 0 };  // 1.. = position in csrc/vsmpc_horizons.def

int select_variant(int n_iter, int n_iter_small, int control_horizon);
int num_variants();
void variant_horizon(int variant, int* n_iter, int* n_iter_small, int* control_horizon);
const char* variant_kernel_name(int variant);
int variant_condensed_dim(int variant);
size_t variant_lds_bytes(int variant);  // dynamic LDS of one workgroup (<= 80 KB: two workgroups share a CU)

int initial_kernel_form();               // VSMPC_FORM: what a new handle starts with
bool variant_has_structured(int variant);

// form: 0 the horizon's default (structured condensing where available), 1 structured, 2 SYRK
hipError_t launch_solve(int variant, int form, const DevCfg& cfg, const double* d_in, int batch, double* d_x, double* d_fm,
                        int* d_status, int* d_iters, double* dbgM, double* dbgL, unsigned long long* stamps,
                        hipStream_t stream);
hipError_t launch_linearize(int variant, const DevCfg& cfg, const double* d_in, int batch, double* A, double* Bj,
                            double* Bt, double* c, hipStream_t stream);

struct KinOpts {
    int sel[VSMPC_N_JOINTS];   // robot joint index of every controlled joint (Lambda_ang columns)
    int constant_lambda;       // jointsLambdaOption "constant"
    double* records = nullptr; // when set: LLIN | LANG | INERTIA are written into these device-resident input records
    int n_in = 0;
    int skip_inertia = 0;      // patching only: leave INERTIA alone (the rollout's tree plant writes R I_B R^T itself)
};
hipError_t launch_kinematics(const double* d_kin, int batch, double* d_out, const KinOpts& opts, hipStream_t stream);

// kinematics provider on a simplified tree (vsmpc_provider.hip)
hipError_t launch_provider(const vsmpc_tree& tree, const double* d_state, int batch, double* d_kin, double* d_robot,
                           double* d_records, int n_in, hipStream_t stream);
// kinematics terms written straight into device-resident input records (LLIN, LANG, INERTIA)
hipError_t launch_kinematics_patch(const double* d_kin, int batch, double* d_records, int n_in, const KinOpts& opts,
                                   hipStream_t stream);

// closed-loop rollout (vsmpc_rollout.hip)
struct RolloutDev {
    int n_in, n_ref, ratio, n_traj, n_alpha;
    int n_ts;         // doubles of per-instance tick state (reference window, RPY unwrap: see vsmpc_rollout.hip)
    int alpha_up;     // up-sampling factor of the alpha-gravity track (TrajectoryManager.cpp:23-39)
    double period_mpc, alpha_dt;
    const double* traj_rpy;      // device, [n_traj][3] or nullptr: RPY / RPYDot tracks of the position trajectory
    const double* traj_rpyd;     //   (vsmpc_rollout_set_attitude_tracks; the shipped ones are all zero)
    // jet plant option (vsmpc_rollout_set_jet_plant): LSTM thrust dynamics + EKF estimates instead of the polynomial model
    // kinematic-tree plant (vsmpc_rollout_set_tree): A_mom,body(q), I_B(q) and the Lambda terms come from the kinematics
    // provider evaluated on the plant's own joint state in the body frame (base at the origin, identity attitude), through
    // these per-instance arrays: Robot-level outputs [batch][VSMPC_RO_SIZE], kinematics terms [batch][VSMPC_KIN_OUT] (I_G
    // = I_B there), and the kinematics record [batch][VSMPC_KIN_SIZE] whose thrusts the advance kernel refreshes
    int tree;
    const double* tree_ro;
    const double* tree_kout;
    double* tree_kin;
    int jet_nn, jet_hidden;
    const float* jet_w;          // device: wih col 0 [4H] | wih col 1 [4H] | b_ih [4H] | b_hh [4H] | fc_w [H] | fc_b
    double jet_norm[4];          // thrust mean / std, throttle mean / std of the checkpoint
    double ekf_q[4], ekf_r[4];
};
// view of a jet handle for the rollout (vsmpc_jet.hip)
void jet_plant_view(const ::vsmpc_jet* j, const float** w, int* hidden, double norm[4], int* device);
struct RolloutCtl {   // device-resident per-run control block of the rollout
    double* log;      // [ticks of this run][batch][VSMPC_ROLLOUT_LOG] or nullptr
    int tick_base;    // tick counter at the start of the run
    int log_rows;     // rows the log buffer holds
};
// provider state of every instance at the body-frame pose: joints = plant joints (+ the first move's increments when `fm`
// is given and the status is Solved: what the advance kernel is about to apply), thrusts as the controller sees them
hipError_t launch_tree_state(const RolloutDev& rd, int batch, const double* state, const double* fm, const int* status,
                             double* rs, hipStream_t stream);
hipError_t launch_record(const RolloutDev& rd, int batch, const double* state, const double* params, const int* tick,
                         const double* traj_pos, const double* traj_vel, const double* traj_alpha, double* tstate,
                         double* rec, hipStream_t stream);
hipError_t launch_advance(const RolloutDev& rd, int batch, double* state, const double* params, int* tick, const double* fm,
                          const int* status, const int* iters, const double* traj_alpha, const RolloutCtl* ctl, int substeps,
                          const double* traj_pos, const double* traj_vel, double* tstate, double* rec_next,
                          hipStream_t stream);

}  // namespace vsmpc
