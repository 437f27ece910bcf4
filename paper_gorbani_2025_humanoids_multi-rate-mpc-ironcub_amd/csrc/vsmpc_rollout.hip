// Closed-loop batched rollout: the reference's hidden per-tick "advance" on the device.
//
//   record_kernel   builds the per-tick input record of every instance from its plant state, i.e. what
//                   IMPCProblem::update pulls out of Robot/QPInput plus the tick state machine, with the reference's own
//                   call-count semantics (k = number of update() calls before this one; configure() made call 0 of every
//                   plugin, IMPCProblem.cpp:94-96):
//                     measured state X0, RPY unwrapped with      constraintsVSMPC.cpp:206-247
//                       per-instance turn counters
//                     20-tick throttle hold: v0 free on ticks    constraintsVSMPC.cpp:335,351-372
//                       k % 20 == 19
//                     reference window = FIFO of 11 columns,     costsVSMPC.cpp:103-113,121-165
//                       initially all trajectory sample 0; a
//                       new sample enters the LAST column in
//                       configure and on ticks k % 20 == 19 (the
//                       throttle-release tick); its h_lin entry
//                       R^T m v is frozen with the R of the push
//                     alpha-gravity cursor: tick k uses sample   systemDynamicsVSMPC.cpp:308-311,
//                       k + 1 of the x20 linearly up-sampled       TrajectoryManager.cpp:23-39,142-153
//                       track (configure consumed sample 0)
//                     joint posture error                        costsVSMPC.cpp:574-589
//   solve_kernel    (vsmpc_kernels.hip) the MPC solve
//   advance_kernel  consumes the first move only if the status is Solved (variableSamplingMPC.cpp:91-108):
//                   q += dq, throttle / thrust references latched; then integrates a centroidal + jet plant over one
//                   MPC period in 1 ms sub-steps (the harness steps MuJoCo 5 x 1 ms per tick,
//                   src/mujoco_lib/ironcub_mujoco_simulator.py:122-139).
//
// The plant is this repo's synthetic stand-in for the MuJoCo/LSTM simulation (out of scope): the
// same centroidal-momentum model the MPC linearises, but with the non-linear rotation kinematics, the joint-dependent
// (bilinear) jet map A_mom(q) T and the non-linear polynomial jet model (utils/src/JetModel.cpp:29-64), integrated
// explicitly.
#include "vsmpc_device.hpp"
#include "vsmpc_jet_device.hpp"
#include "vsmpc_launch.hpp"

namespace vsmpc {

VS_DEV void rot_from_rpy(const double* rpy, double* R) {  // Rz(yaw) Ry(pitch) Rx(roll), row-major
    double sr, cr, sp, cp, sy, cy;
    sincos(rpy[0], &sr, &cr); sincos(rpy[1], &sp, &cp); sincos(rpy[2], &sy, &cy);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

VS_DEV void inv3(const double* I, double* Ii) {
    const double a = I[0], b = I[1], c = I[2], d = I[3], e = I[4], f = I[5], g = I[6], h = I[7], k = I[8];
    const double A00 = e * k - f * h, A01 = c * h - b * k, A02 = b * f - c * e;
    const double A10 = f * g - d * k, A11 = a * k - c * g, A12 = c * d - a * f;
    const double A20 = d * h - e * g, A21 = b * g - a * h, A22 = a * e - b * d;
    const double idet = 1.0 / (a * A00 + b * A10 + c * A20);
    Ii[0] = A00 * idet; Ii[1] = A01 * idet; Ii[2] = A02 * idet; Ii[3] = A10 * idet; Ii[4] = A11 * idet;
    Ii[5] = A12 * idet; Ii[6] = A20 * idet; Ii[7] = A21 * idet; Ii[8] = A22 * idet;
}

VS_DEV double interp_clamped(const double* __restrict__ tr, int n, double pos) {  // linear up-sampling, clamped
    if (pos <= 0.0) return tr[0];
    if (pos >= double(n - 1)) return tr[n - 1];
    const int i = int(pos);
    const double f = pos - double(i);
    return tr[i] + f * (tr[i + 1] - tr[i]);
}

// One wavefront per instance.  The plant state and parameters are staged through LDS with coalesced loads, the record
// is assembled in LDS by all 64 lanes (every field is a short independent expression) and written out coalesced.
constexpr int RO_BLOCK = 64;

// global -> LDS staging with all loads in flight before the first store (a load/store loop pays one HBM round trip
// per iteration)
template <int N>
VS_DEV void stage_in(const double* __restrict__ g, double* __restrict__ l, int lane) {
    constexpr int R = (N + RO_BLOCK - 1) / RO_BLOCK;
    double v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = lane + k * RO_BLOCK;
        v[k] = i < N ? g[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = lane + k * RO_BLOCK;
        if (i < N) l[i] = v[k];
    }
}

// Per-instance tick state that survives between ticks, VSMPC tick-state record in HBM:
//   [0, 12 n_ref)   reference window, column-major: col * 12 + (p 0:3 | h_lin 3:6 | rpy 6:9 | h_ang 9:12)
//   + 0..2          m_rpyOld (last measured, wrapped RPY)      constraintsVSMPC.cpp:246
//   + 3..5          m_nTurns                                    constraintsVSMPC.cpp:236-243
VS_HD constexpr int tick_state_doubles(int n_ref) { return 12 * n_ref + 8; }

VS_DEV double wrap_pi(double a) {  // what Rotation::asRPY() returns for an angle the plant integrates continuously
    const double two_pi = 6.283185307179586476925286766559;
    return a - two_pi * rint(a / two_pi);
}

// TrajectoryManager semantics for alpha-gravity: linear up-sampling by `up` (TrajectoryManager.cpp:23-39: the last
// original sample is dropped), cursor advanced once per dynamics evaluation and clamped at the last up-sampled sample
// (:142-153); configure consumed sample 0, so tick k reads sample k + 1.
VS_DEV double alpha_of_tick(const double* __restrict__ tr, int n, int up, int tk) {
    // (a track already at the requested rate is NOT resampled and keeps all n samples, TrajectoryManager.cpp:121-126)
    const int last = up == 1 ? n - 1 : up * (n - 1) - 1;
    int idx = tk + 1;
    idx = idx < last ? idx : last;
    idx = idx > 0 ? idx : 0;
    const int i = idx / up, j = idx - up * i;
    const double v0 = tr[i], v1 = tr[i + 1 < n ? i + 1 : n - 1];
    return v0 + (v1 - v0) * (double(j) / double(up));
}

// Record of tick `tk` from the plant state `s`, the parameters `p` and the tick state `ts` (all in LDS), assembled into
// `r` (LDS) by the 64 lanes of the workgroup; `ts` is advanced to "after update() number tk".  `first` builds the tick
// state itself: what configure() leaves behind, fast-forwarded to tick `tk` for loops that start mid-trajectory (window
// columns frozen with the current R, as if the attitude had been constant before).  Ends with a barrier.
// Tree plant (rd.tree): IB = I_B(q) and amom = A_mom,body(q) of the kinematic tree (LDS copies of the provider's outputs for
// this instance) take the place of the plant parameters; Lambda_lin / Lambda_ang are left to the kinematics kernel, which
// patches them into the finished record from the tree's Jacobians and the thrusts the advance kernel measured.
VS_DEV void assemble_record(const RolloutDev& rd, const double* __restrict__ s, const double* __restrict__ p, int tk,
                            const double* __restrict__ traj_pos, const double* __restrict__ traj_vel,
                            const double* __restrict__ traj_alpha, double* __restrict__ R, double* __restrict__ om,
                            double* __restrict__ ts, double* __restrict__ r, int lane, bool first,
                            const double* __restrict__ IB, const double* __restrict__ amom) {
    const double m = p[VSMPC_PP_MASS];
    const int nref = rd.n_ref, nwin = 12 * rd.n_ref;
    double* rpy_old = ts + nwin;
    double* n_turns = ts + nwin + 3;
    const double two_pi = 6.283185307179586476925286766559;
    if (lane == 0) {
        rot_from_rpy(s + VSMPC_PS_RPY, R);
    } else if (lane == 1) {
        double IBi[9];
        inv3(IB, IBi);
        for (int i = 0; i < 3; ++i)
            om[i] = IBi[3 * i] * s[VSMPC_PS_HANG] + IBi[3 * i + 1] * s[VSMPC_PS_HANG + 1] + IBi[3 * i + 2] * s[VSMPC_PS_HANG + 2];
    } else if (lane >= 2 && lane < 5) {
        // ConstraintInitialState::unwrapRPY (constraintsVSMPC.cpp:232-247) on the wrapped measurement
        const int i = lane - 2;
        const double meas = wrap_pi(s[VSMPC_PS_RPY + i]);
        if (first) {
            n_turns[i] = rint((s[VSMPC_PS_RPY + i] - meas) / two_pi);   // m_rpyOld = initial RPY, turns as flown so far
        } else {
            const double d = meas - rpy_old[i];
            if (d > 3.14159265358979323846) n_turns[i] -= 1.0;
            else if (d < -3.14159265358979323846) n_turns[i] += 1.0;
        }
        rpy_old[i] = meas;
    }
    __syncthreads();
    // window column from trajectory sample `idx` with the CURRENT attitude and mass (costsVSMPC.cpp:103-113,127-146)
    auto column_entry = [&](int idx, int i) -> double {
        idx = idx < rd.n_traj ? idx : rd.n_traj - 1;
        if (i < 3) return p[VSMPC_PP_PINIT + i] + traj_pos[3 * idx + i];   // m_initialCoMPos + positionCoM
        if (i < 6) {                                                        // R^T m v_ref
            const int c = i - 3;
            return R[c] * (m * traj_vel[3 * idx]) + R[3 + c] * (m * traj_vel[3 * idx + 1]) + R[6 + c] * (m * traj_vel[3 * idx + 2]);
        }
        if (i < 9)                                                          // m_initialRPY + RPY trajectory
            return p[VSMPC_PP_RPYINIT + i - 6] + (rd.traj_rpy != nullptr ? rd.traj_rpy[3 * idx + i - 6] : 0.0);
        if (rd.traj_rpyd == nullptr) return 0.0;                            // the shipped RPYDot track is all zero (A.6)
        // m_inertia * m_W * RPYDot with the CURRENT attitude (costsVSMPC.cpp:111-112,143-146,266-286); the locked inertia
        // of this plant is R I_B R^T
        const double r0 = s[VSMPC_PS_RPY], p0 = s[VSMPC_PS_RPY + 1];
        const double d0 = rd.traj_rpyd[3 * idx], d1 = rd.traj_rpyd[3 * idx + 1], d2 = rd.traj_rpyd[3 * idx + 2];
        const double wd[3] = {d0 - sin(p0) * d2, cos(r0) * d1 + cos(p0) * sin(r0) * d2, -sin(r0) * d1 + cos(r0) * cos(p0) * d2};
        double acc = 0.0;
        const int row = i - 9;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                double rw = 0.0;
                for (int j = 0; j < 3; ++j) rw += R[3 * j + c] * wd[j];     // (R^T wd)[c]
                acc += R[3 * row + a] * IB[3 * a + c] * rw;
            }
        return acc;
    };
    if (first) {
        // shifts made before update() number tk: the one of configure + one per earlier tick with k % ratio == ratio - 1;
        // shift number s pushed trajectory sample min(s, n_traj - 1), the initial fill is sample 0
        const int ns0 = 1 + tk / rd.ratio;
        for (int e = lane; e < nwin; e += RO_BLOCK) {
            const int j = e / 12, i = e - 12 * j;
            const int sj = ns0 - (nref - 1 - j);
            ts[e] = column_entry(sj > 0 ? sj : 0, i);
        }
        __syncthreads();
    }
    if (tk % rd.ratio == rd.ratio - 1) {   // ReferenceTrackingCost::computeHessianAndGradient, m_counter == ratio - 1
        const int ns = 1 + (tk + 1) / rd.ratio;
        constexpr int KMAX = (12 * MAX_STAGES + RO_BLOCK - 1) / RO_BLOCK;
        double keep[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + k * RO_BLOCK;
            const int j = e / 12, i = e - 12 * j;
            keep[k] = e < nwin ? (j + 1 < nref ? ts[e + 12] : column_entry(ns, i)) : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + k * RO_BLOCK;
            if (e < nwin) ts[e] = keep[k];
        }
        __syncthreads();
    }
    for (int e = lane; e < nwin; e += RO_BLOCK) r[VSMPC_IN_XREF + e] = ts[e];
    // A_mom(q) = A_mom0 + sum_j DJ[j] (q_j - q_ref0_j); Lambda column j = DJ[j] T at the measured thrust
    // (the reference recomputes both from the kinematics each tick, systemDynamicsVSMPC.cpp:159-206,304,321-350)
    if (lane < 24) {
        double acc = p[VSMPC_PP_AMOM0 + lane];
        for (int j = 0; j < 8; ++j) acc += p[VSMPC_PP_DJ + 24 * j + lane] * (s[VSMPC_PS_Q + j] - p[VSMPC_PP_QREF0 + j]);
        r[VSMPC_IN_AMOM + lane] = rd.tree ? amom[lane] : acc;
    }
    // thrusts and thrust rates as the controller sees them: the plant's own with the polynomial jet plant, the EKF
    // estimates with the NN jet plant (the harness hands the estimate to Robot::setJetThrusts)
    auto meas_T = [&](int i) { return rd.jet_nn ? s[VSMPC_PS_EST + 2 * i] : s[VSMPC_PS_T + i]; };
    auto meas_Td = [&](int i) { return rd.jet_nn ? s[VSMPC_PS_EST + 2 * i + 1] : s[VSMPC_PS_TD + i]; };
    if (lane < 48) {
        const int row = lane >> 3, j = lane & 7;   // row 0..5 of DJ[j] T
        double acc = 0.0;
        for (int c = 0; c < 4; ++c) acc += p[VSMPC_PP_DJ + 24 * j + 4 * row + c] * meas_T(c);
        if (rd.tree) acc = 0.0;   // (patched in afterwards by the kinematics kernel)
        if (row < 3) r[VSMPC_IN_LLIN + 8 * row + j] = acc;
        else r[VSMPC_IN_LANG + 8 * (row - 3) + j] = acc;
    }
    if (lane >= 48 && lane < 57) {                  // I_G = R I_B R^T (world-oriented locked inertia)
        const int i = (lane - 48) / 3, j = (lane - 48) % 3;
        double acc = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) acc += R[3 * i + a] * IB[3 * a + c] * R[3 * j + c];
        r[VSMPC_IN_INERTIA + 3 * i + j] = acc;
    }
    // X0 (constraintsVSMPC.cpp:206-230): RPY enters unwrapped
    if (lane < 20)
        r[VSMPC_IN_X0 + lane] = (lane >= 6 && lane < 9) ? rpy_old[lane - 6] + two_pi * n_turns[lane - 6]
                                : (lane >= 16 ? meas_Td(lane - 16) : (lane >= 12 ? meas_T(lane - 12) : s[lane]));
    if (lane >= 20 && lane < 29) r[VSMPC_IN_WRB + lane - 20] = R[lane - 20];
    if (lane >= 32 && lane < 36) {
        const int i = lane - 32;
        r[VSMPC_IN_T0 + i] = meas_T(i);                  // useEstimatedThrust = true (systemDynamicsVSMPC.cpp:401-404)
        r[VSMPC_IN_TD0 + i] = meas_Td(i);
        r[VSMPC_IN_UPREV + i] = s[VSMPC_PS_U + i];
        r[VSMPC_IN_TDES + i] = s[VSMPC_PS_TDES + i];
        r[VSMPC_IN_TDDES + i] = s[VSMPC_PS_TDDES + i];
    }
    if (lane >= 36 && lane < 44) r[VSMPC_IN_QERR + lane - 36] = s[VSMPC_PS_Q + lane - 36] - p[VSMPC_PP_QREF0 + lane - 36];
    if (lane >= 44 && lane < 47) {
        const int i = lane - 44;
        r[VSMPC_IN_OMEGA + i] = om[i];
        r[VSMPC_IN_RPY + i] = rpy_old[i];                // Rotation::asRPY() of the measurement (systemDynamicsVSMPC.cpp:132)
        r[VSMPC_IN_RPYINIT + i] = p[VSMPC_PP_RPYINIT + i];
        r[VSMPC_IN_GRAV + i] = i == 2 ? -9.81 : 0.0;
    }
    if (lane == 47) {
        r[VSMPC_IN_MASS] = m;
        r[VSMPC_IN_ALPHA] = alpha_of_tick(traj_alpha, rd.n_alpha, rd.alpha_up, tk);
        r[VSMPC_IN_HOLD] = (tk % rd.ratio) != (rd.ratio - 1) ? 1.0 : 0.0;   // constraintsVSMPC.cpp:351,366-372
    }
    __syncthreads();
    if (lane < 3) {                                  // QPInput::getPosCoMReference / getRPYReference = column 0 (costsVSMPC.cpp:155-156)
        r[VSMPC_IN_X0 + 20 + lane] = s[VSMPC_PS_P + lane] - r[VSMPC_IN_XREF + lane];
        r[VSMPC_IN_X0 + 23 + lane] = r[VSMPC_IN_X0 + 6 + lane] - r[VSMPC_IN_XREF + 6 + lane];
        r[VSMPC_IN_PREF + lane] = r[VSMPC_IN_XREF + lane];
    }
    __syncthreads();
}

__global__ __launch_bounds__(RO_BLOCK) void record_kernel(RolloutDev rd, int batch, const double* __restrict__ state,
                                                           const double* __restrict__ params, const int* __restrict__ tick,
                                                           const double* __restrict__ traj_pos, const double* __restrict__ traj_vel,
                                                           const double* __restrict__ traj_alpha, double* __restrict__ tstate,
                                                           double* __restrict__ rec) {
    __shared__ double s[VSMPC_PLANT_STATE], p[VSMPC_PLANT_PARAMS + 1], R[9], om[3], tIB[9], tAm[24];
    __shared__ double r[VSMPC_IN_XREF + 12 * MAX_STAGES], ts[12 * MAX_STAGES + 8];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= batch) return;
    stage_in<VSMPC_PLANT_STATE>(state + size_t(b) * VSMPC_PLANT_STATE, s, lane);
    stage_in<VSMPC_PLANT_PARAMS>(params + size_t(b) * VSMPC_PLANT_PARAMS, p, lane);
    if (rd.tree) {
        if (lane < 9) tIB[lane] = rd.tree_kout[size_t(b) * VSMPC_KIN_OUT + 48 + lane];
        if (lane >= 32 && lane < 56) tAm[lane - 32] = rd.tree_ro[size_t(b) * VSMPC_RO_SIZE + VSMPC_RO_AMOMB + lane - 32];
    }
    __syncthreads();
    assemble_record(rd, s, p, tick[b] + int(p[VSMPC_PP_TICK0]), traj_pos, traj_vel, traj_alpha, R, om, ts, r, lane, true,
                    rd.tree ? tIB : p + VSMPC_PP_INERTIA_B, tAm);
    double* out = rec + size_t(b) * rd.n_in;
    for (int e = lane; e < rd.n_in; e += RO_BLOCK) out[e] = r[e];
    double* tso = tstate + size_t(b) * rd.n_ts;
    for (int e = lane; e < rd.n_ts; e += RO_BLOCK) tso[e] = ts[e];
}

// One wavefront per instance: staging and the joint-dependent jet map in parallel, the short ODE integration in lane 0
// on registers, coalesced write-back.  With `rec_next` the record of the NEXT tick is assembled from the advanced state
// while it is still in LDS (one launch and one staging less per tick than a separate record_kernel).
__global__ __launch_bounds__(RO_BLOCK) void advance_kernel(RolloutDev rd, int batch, double* __restrict__ state,
                                                            const double* __restrict__ params, int* __restrict__ tick,
                                                            const double* __restrict__ fm, const int* __restrict__ status,
                                                            const int* __restrict__ iters, const double* __restrict__ traj_alpha,
                                                            const RolloutCtl* __restrict__ ctl, int substeps,
                                                            const double* __restrict__ traj_pos,
                                                            const double* __restrict__ traj_vel, double* __restrict__ tstate,
                                                            double* __restrict__ rec_next) {
    __shared__ double s[VSMPC_PLANT_STATE], p[VSMPC_PLANT_PARAMS + 1], f[VSMPC_FM_SIZE], Aq[24], IBi[9], R[9], om[3], tIB[9];
    __shared__ double r[VSMPC_IN_XREF + 12 * MAX_STAGES], ts[12 * MAX_STAGES + 8];
    __shared__ float jw[17 * JET_HMAX + 1];   // LSTM weights of the jet plant option
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= batch) return;
    if (rd.jet_nn)
        for (int e = lane; e < 17 * rd.jet_hidden + 1; e += RO_BLOCK) jw[e] = rd.jet_w[e];
    for (int e = lane; e < rd.n_ts; e += RO_BLOCK) ts[e] = tstate[size_t(b) * rd.n_ts + e];
    stage_in<VSMPC_PLANT_STATE>(state + size_t(b) * VSMPC_PLANT_STATE, s, lane);
    stage_in<VSMPC_PLANT_PARAMS>(params + size_t(b) * VSMPC_PLANT_PARAMS, p, lane);
    stage_in<VSMPC_FM_SIZE>(fm + size_t(b) * VSMPC_FM_SIZE, f, lane);
    if (rd.tree && lane < 9) tIB[lane] = rd.tree_kout[size_t(b) * VSMPC_KIN_OUT + 48 + lane];   // I_B(q) of the tree, q = the joints
    double tree_amom = 0.0;                                                                     // AFTER this tick's move
    if (rd.tree && lane < 24) tree_amom = rd.tree_ro[size_t(b) * VSMPC_RO_SIZE + VSMPC_RO_AMOMB + lane];
    const int st = status[b];
    const int tick_before = tick[b];
    // every global load of this kernel is requested as early as possible: each dependent round trip costs ~1.5 us
    double* const log = ctl->log;
    const int tick_base = ctl->tick_base, log_rows = ctl->log_rows;
    __syncthreads();
    __shared__ double alpha_s[16];   // alpha-gravity of every sub-step (one round trip instead of one per sub-step)
    if (lane >= 32 && lane < 32 + substeps && lane < 48) {
        const int ss = lane - 32;
        const double t = (double(tick_before + int(p[VSMPC_PP_TICK0])) + double(ss) / double(substeps)) * rd.period_mpc;
        alpha_s[ss] = interp_clamped(traj_alpha, rd.n_alpha, t / rd.alpha_dt);
    }
    if (st == VSMPC_STATUS_SOLVED) {  // variableSamplingMPC.cpp:91-108
        if (lane < 8) s[VSMPC_PS_Q + lane] += f[VSMPC_FM_DQ + lane];
        if (lane >= 8 && lane < 12) {
            const int i = lane - 8;
            s[VSMPC_PS_U + i] = f[VSMPC_FM_THROTTLE + i];
            s[VSMPC_PS_TDES + i] = f[VSMPC_FM_THRUST + i];
            s[VSMPC_PS_TDDES + i] = f[VSMPC_FM_THRUSTDOT + i];
        }
    }
    __syncthreads();
    if (lane < 24) {  // A_mom(q): the joints only move at the tick boundary
        double acc = p[VSMPC_PP_AMOM0 + lane];
        for (int j = 0; j < 8; ++j) acc += p[VSMPC_PP_DJ + 24 * j + lane] * (s[VSMPC_PS_Q + j] - p[VSMPC_PP_QREF0 + j]);
        Aq[lane] = rd.tree ? tree_amom : acc;
    } else if (lane == 24) {
        inv3(rd.tree ? tIB : p + VSMPC_PP_INERTIA_B, IBi);
    }
    __syncthreads();
    // explicit Euler sub-steps, one state component per lane (20 lanes): the three sincos run side by side in lanes
    // 0..2, the body rates in lanes 8..10, then every lane forms the derivative of its own component
    {
        __shared__ double x[20], Rm[9], omv[3], sc[6], vthr[4];
        const int tk = tick_before + int(p[VSMPC_PP_TICK0]);
        const double m = p[VSMPC_PP_MASS];
        if (lane < 20) x[lane] = s[lane];
        if (lane < 4) vthr[lane] = Jet::v_of_throttle(s[VSMPC_PS_U + lane]);
        __syncthreads();
        const double h = rd.period_mpc / double(substeps);
        for (int ss = 0; ss < substeps; ++ss) {
            const double t = (double(tk) + double(ss) / double(substeps)) * rd.period_mpc;
            const double alpha = alpha_s[ss];
            const bool dist = t >= p[VSMPC_PP_DIST_T0] && t < p[VSMPC_PP_DIST_T1];
            if (rd.jet_nn) {
                // jet plant option, one plant step in the order of MujocoSim.step (ironcub_mujoco_simulator.py:128-133,
                // 393-396): the NN advances every jet's thrust from its own previous output (hidden units over the lanes),
                // the jet's EKF takes the NN's (T, Tdot) as measurement, set_thrust(estimated_thrust) -- and only then the
                // body is stepped: the force of THIS sub-step is the EKF estimate of this sub-step.
                const JetNorm nm{rd.jet_norm[0], rd.jet_norm[1], rd.jet_norm[2], rd.jet_norm[3]};
                float Tn_mine = 0.0f, Tdn_mine = 0.0f;
                for (int j = 0; j < 4; ++j) {
                    float x0, x1;
                    jet_normalize(nm, float(s[VSMPC_PS_TNN + j]), float(s[VSMPC_PS_U + j]), x0, x1);
                    const float out = lstm_step_from_zero_wave(jw, rd.jet_hidden, x0, x1, lane);
                    const float Tn = (x0 + out * float(h)) * float(nm.thrust_std) + float(nm.thrust_mean);
                    const float Tdn = out * float(nm.thrust_std);
                    if (lane == j) { Tn_mine = Tn; Tdn_mine = Tdn; }
                }
                __syncthreads();                              // every lane has read s[TNN] before lanes 0..3 update it
                if (lane < 4) {
                    Ekf2 cv;
                    for (int k = 0; k < 4; ++k) { cv.q[k] = rd.ekf_q[k]; cv.r[k] = rd.ekf_r[k]; }
                    double T = s[VSMPC_PS_EST + 2 * lane], Td = s[VSMPC_PS_EST + 2 * lane + 1];
                    double Pm[4] = {s[VSMPC_PS_EKFP + 4 * lane], s[VSMPC_PS_EKFP + 4 * lane + 1],
                                    s[VSMPC_PS_EKFP + 4 * lane + 2], s[VSMPC_PS_EKFP + 4 * lane + 3]};
                    ekf_update_dev(T, Td, Pm, s[VSMPC_PS_U + lane], double(Tn_mine), double(Tdn_mine), h, cv);
                    s[VSMPC_PS_EST + 2 * lane] = T;
                    s[VSMPC_PS_EST + 2 * lane + 1] = Td;
                    for (int k = 0; k < 4; ++k) s[VSMPC_PS_EKFP + 4 * lane + k] = Pm[k];
                    s[VSMPC_PS_TNN + lane] = double(Tn_mine);   // the NN's own feedback state
                    x[12 + lane] = T;                           // set_thrust(self._estimated_thrust)
                    x[16 + lane] = Td;
                }
                __syncthreads();
            }
            if (lane < 3) {
                double sn, cs;
                sincos(x[6 + lane], &sn, &cs);
                sc[2 * lane] = sn;
                sc[2 * lane + 1] = cs;
            } else if (lane >= 8 && lane < 11) {
                const int i = lane - 8;
                omv[i] = IBi[3 * i] * x[9] + IBi[3 * i + 1] * x[10] + IBi[3 * i + 2] * x[11];
            }
            __syncthreads();
            const double sr = sc[0], cr = sc[1], sp = sc[2], cp = sc[3], sy = sc[4], cy = sc[5];
            if (lane == 0) {
                Rm[0] = cy * cp; Rm[1] = cy * sp * sr - sy * cr; Rm[2] = cy * sp * cr + sy * sr;
                Rm[3] = sy * cp; Rm[4] = sy * sp * sr + cy * cr; Rm[5] = sy * sp * cr - cy * sr;
                Rm[6] = -sp;     Rm[7] = cp * sr;                Rm[8] = cp * cr;
            }
            __syncthreads();
            double d = 0.0;
            if (lane < 3) {                                   // p' = R h_lin / m
                d = (Rm[3 * lane] * x[3] + Rm[3 * lane + 1] * x[4] + Rm[3 * lane + 2] * x[5]) / m;
            } else if (lane < 6 || (lane >= 9 && lane < 12)) { // momentum rates: -omega x h + A_mom(q) T (+ gravity, push)
                const bool lin = lane < 6;
                const int r = lin ? lane - 3 : lane - 9, h0 = lin ? 3 : 9;
                const int r1 = (r + 1) % 3, r2 = (r + 2) % 3;
                const double cross = omv[r1] * x[h0 + r2] - omv[r2] * x[h0 + r1];   // (omega x h)[r]
                double f = -cross;
                if (lin) f += alpha * m * (Rm[6 + r] * (-9.81));                    // alpha m R^T g, g = (0,0,-9.81)
                for (int j = 0; j < 4; ++j) f += Aq[4 * (lin ? r : 3 + r) + j] * x[12 + j];
                if (dist)
                    f += lin ? Rm[r] * p[VSMPC_PP_DIST_F] + Rm[3 + r] * p[VSMPC_PP_DIST_F + 1] + Rm[6 + r] * p[VSMPC_PP_DIST_F + 2]
                             : p[VSMPC_PP_DIST_TAU + r];
                d = f;
            } else if (lane < 9) {                            // rpy' = W^-1 omega (systemDynamicsVSMPC.cpp:140-147)
                const double tp = sp / cp;
                d = lane == 6 ? omv[0] + sr * tp * omv[1] + cr * tp * omv[2]
                  : lane == 7 ? cr * omv[1] - sr * omv[2]
                              : (sr * omv[1] + cr * omv[2]) / cp;
            } else if (rd.jet_nn) {                           // thrusts follow the NN below, not an ODE
                d = 0.0;
            } else if (lane < 16) {                           // T' = Tdot
                d = x[lane + 4];
            } else if (lane < 20) {                           // T'' = sigma_T (f + g v(u))  (JetModel.cpp:29-64)
                const int j = lane - 16;
                const double Tb = Jet::stdT(x[12 + j]), Tdb = Jet::stdTd(x[16 + j]);
                d = Jet::sgT * (Jet::f(Tb, Tdb) + Jet::g(Tb, Tdb) * vthr[j]);
            }
            __syncthreads();                                  // every lane has read x before anyone updates it
            if (lane < 20) x[lane] += h * d;
            __syncthreads();
        }
        if (lane < 20) s[lane] = x[lane];
        if (lane == 0) tick[b] = tick_before + 1;
    }
    __syncthreads();
    double* so = state + size_t(b) * VSMPC_PLANT_STATE;
    for (int i = lane; i < VSMPC_PLANT_STATE; i += RO_BLOCK) so[i] = s[i];
    // the log destination comes from a device-side control block, so the launch arguments are the same for every
    // tick and a captured graph of ticks can be replayed
    if (log != nullptr && lane < VSMPC_ROLLOUT_LOG) {
        const int row = tick_before - tick_base;   // ticks since the start of this run
        double v;
        if (lane < 3) v = s[lane];
        else if (lane < 6) v = s[6 + lane - 3];
        else if (lane < 10) v = s[12 + lane - 6];
        else if (lane < 14) v = s[VSMPC_PS_U + lane - 10];
        else if (lane == 14) v = double(st);
        else v = iters ? double(iters[b]) : 0.0;
        // a run that failed half-way leaves the counters ahead: never write outside the log (and never leave the
        // workgroup here: the record of the next tick is assembled below, with barriers)
        if (row >= 0 && row < log_rows) log[(size_t(row) * batch + b) * VSMPC_ROLLOUT_LOG + lane] = v;
    }
    if (rd.tree && lane < 4)   // the thrusts the next tick's Lambda terms are formed with (kinematics kernel, next launch)
        rd.tree_kin[size_t(b) * VSMPC_KIN_SIZE + VSMPC_KIN_THRUST + lane] = rd.jet_nn ? s[VSMPC_PS_EST + 2 * lane] : s[VSMPC_PS_T + lane];
    if (rec_next != nullptr) {
        assemble_record(rd, s, p, tick_before + 1 + int(p[VSMPC_PP_TICK0]), traj_pos, traj_vel, traj_alpha, R, om, ts, r, lane, false,
                        rd.tree ? tIB : p + VSMPC_PP_INERTIA_B, Aq);
        double* out = rec_next + size_t(b) * rd.n_in;
        for (int e = lane; e < rd.n_in; e += RO_BLOCK) out[e] = r[e];
        double* tso = tstate + size_t(b) * rd.n_ts;
        for (int e = lane; e < rd.n_ts; e += RO_BLOCK) tso[e] = ts[e];
    }
}

// Provider state of the tree plant, one thread per instance: the base frame is the body frame (origin, identity attitude,
// at rest), so that every output of the provider is a body-frame quantity of the joints alone -- A_mom,body(q), the
// Jacobians behind Lambda_lin,B / Lambda_ang,B (systemDynamicsVSMPC.cpp:159-226,321-350 are body-frame expressions) and
// I_B(q).  The joints are the plant's, plus this tick's increments when the solve succeeded (variableSamplingMPC.cpp:
// 104-108): the advance kernel applies the same move and integrates with the A_mom and I_B of the moved joints.
__global__ void tree_state_kernel(RolloutDev rd, int batch, const double* __restrict__ state, const double* __restrict__ fm,
                                  const int* __restrict__ status, double* __restrict__ rs) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double* s = state + size_t(b) * VSMPC_PLANT_STATE;
    double* o = rs + size_t(b) * VSMPC_RS_SIZE;
    for (int i = 0; i < VSMPC_RS_SIZE; ++i) o[i] = 0.0;
    o[VSMPC_RS_R + 0] = 1.0; o[VSMPC_RS_R + 4] = 1.0; o[VSMPC_RS_R + 8] = 1.0;
    const bool move = fm != nullptr && status[b] == VSMPC_STATUS_SOLVED;
    for (int j = 0; j < 8; ++j) o[VSMPC_RS_Q + j] = s[VSMPC_PS_Q + j] + (move ? fm[size_t(b) * VSMPC_FM_SIZE + VSMPC_FM_DQ + j] : 0.0);
    for (int i = 0; i < 4; ++i) o[VSMPC_RS_T + i] = rd.jet_nn ? s[VSMPC_PS_EST + 2 * i] : s[VSMPC_PS_T + i];
}

hipError_t launch_tree_state(const RolloutDev& rd, int batch, const double* state, const double* fm, const int* status,
                             double* rs, hipStream_t stream) {
    hipLaunchKernelGGL(tree_state_kernel, dim3((batch + 63) / 64), dim3(64), 0, stream, rd, batch, state, fm, status, rs);
    return hipGetLastError();
}

hipError_t launch_record(const RolloutDev& rd, int batch, const double* state, const double* params, const int* tick,
                         const double* traj_pos, const double* traj_vel, const double* traj_alpha, double* tstate,
                         double* rec, hipStream_t stream) {
    hipLaunchKernelGGL(record_kernel, dim3(batch), dim3(RO_BLOCK), 0, stream, rd, batch, state, params, tick,
                       traj_pos, traj_vel, traj_alpha, tstate, rec);
    return hipGetLastError();
}

hipError_t launch_advance(const RolloutDev& rd, int batch, double* state, const double* params, int* tick, const double* fm,
                          const int* status, const int* iters, const double* traj_alpha, const RolloutCtl* ctl, int substeps,
                          const double* traj_pos, const double* traj_vel, double* tstate, double* rec_next,
                          hipStream_t stream) {
    hipLaunchKernelGGL(advance_kernel, dim3(batch), dim3(RO_BLOCK), 0, stream, rd, batch, state, params, tick, fm,
                       status, iters, traj_alpha, ctl, substeps, traj_pos, traj_vel, tstate, rec_next);
    return hipGetLastError();
}

}  // namespace vsmpc
