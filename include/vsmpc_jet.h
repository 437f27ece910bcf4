/*
 * vsmpc_jet.h — C-ABI of the batched jet plant / estimator side, MI355X-native.
 *
 * What it replaces in ami-iit/paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub (paths relative to /root/reference/):
 *   src/mujoco_lib/nn_jet_model.py:3-30,86-109     NeuralJetModel (LSTM(2 -> 80) + Linear(80 -> 1), float32) and
 *                                                  JetModelTotal.get_state: one jet at a time, zero initial LSTM state
 *   src/mujoco_lib/jet_kalman_filter.py:4-81       SecondOrderJetModel (casadi) + EKFJetsTotal.update
 *   src/mujoco_lib/ironcub_mujoco_simulator.py:128-133,393-396   the 1 kHz plant step: NN thrust -> EKF -> thrust estimate
 * with a batch axis: a "series" is one jet of one instance (Monte-Carlo closed loops: 4096 instances x 4 jets).
 * Plain C: pointers and sizes only.  Host pointers unless a name says `_device`.  Returns 0 or a VSMPC_ERR_* code
 * (include/vsmpc.h); vsmpc_strerror() explains HIP failures.
 */
#ifndef VSMPC_JET_H
#define VSMPC_JET_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsmpc_jet vsmpc_jet;

/* Weights of NeuralJetModel(2, hidden, 1) exactly as torch stores them (nn_jet_model.py:6-7; gate order i, f, g, o):
 * w_ih[4H][2], w_hh[4H][H], b_ih[4H], b_hh[4H], fc_w[H], fc_b[1]; norm = thrust_mean, thrust_std, throttle_mean,
 * throttle_std of the checkpoint's metadata (:59-62).  hidden <= 128.  Allocates for max_series series. */
int vsmpc_jet_create(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* fc_w,
                     const float* fc_b, const double* norm, int hidden, int device, int max_series, vsmpc_jet** out);
void vsmpc_jet_destroy(vsmpc_jet* j);

/* JetModelTotal.get_state (nn_jet_model.py:86-109) for n series: thrust[n] (N), throttle[n] (percent), dt -> T_next[n],
 * T_dot[n]; h_out / c_out ([n][hidden], the cell state after the step) may be NULL. */
int vsmpc_jet_nn_step(vsmpc_jet* j, const float* thrust, const float* throttle, int n, float dt, float* T_next,
                      float* T_dot, float* h_out, float* c_out);

/* NeuralJetModel.get_state (nn_jet_model.py:16-30) on n sequences of L normalised samples x[n][L][2], state carried
 * through the sequence: T_next_norm[n], T_dot_norm[n], h_n[n][hidden], c_n[n][hidden] (the last two may be NULL). */
int vsmpc_jet_nn_sequence(vsmpc_jet* j, const float* x, int n, int L, float dt, float* T_next_norm, float* T_dot_norm,
                          float* h_n, float* c_n);

/* SecondOrderJetModel.update (jet_kalman_filter.py:57-66) for n series, in place: x[n][2] = (T, T_dot), P[n][4]
 * row-major, throttle u[n], measurement z[n][2]; Q[4], R[4] row-major (ironcub_mujoco_simulator.py:54-56). */
int vsmpc_jet_ekf_update(vsmpc_jet* j, double* x, double* P, const double* u, const double* z, int n, double dt,
                         const double* Q, const double* R);

/* `steps` plant steps of MujocoSim.step with use_nn_jet_dynamics (ironcub_mujoco_simulator.py:128-133), fused, state in
 * registers: per step the NN advances its own thrust T_nn (fed back, :393-396) and every jet's EKF is updated with the
 * NN's (T, T_dot).  In place: T_nn[n] (float), x_est[n][2], P[n][4].  throttle[n] if throttle_steps == 1 (held), else
 * throttle[steps][n].  log (may be NULL) receives the estimates after every step, [steps][n][2]. */
int vsmpc_jet_plant_run(vsmpc_jet* j, float* T_nn, double* x_est, double* P, const float* throttle, int throttle_steps,
                        int n, int steps, double dt, const double* Q, const double* R, double* log);

/* Same on device-resident buffers (all pointers device, no copies, enqueued on `stream`). */
int vsmpc_jet_plant_run_device(vsmpc_jet* j, float* d_T_nn, double* d_x_est, double* d_P, const float* d_throttle,
                               int throttle_steps, int n, int steps, double dt, const double* Q, const double* R,
                               double* d_log, void* stream);

/* Jet plant option of the closed-loop rollout (include/vsmpc.h, vsmpc_rollout_*): every 1 ms plant sub-step runs in the
 * order of MujocoSim.step (ironcub_mujoco_simulator.py:128-133): the LSTM model advances the thrust of each jet (thrust
 * fed back, :393-396), the jet's EKF takes the NN's (T, Tdot) as measurement, set_thrust(estimated_thrust) -- and then
 * the body is stepped, so BOTH the plant's forces and the MPC's records (X0 thrusts and rates, linearisation point,
 * Lambda terms, through Robot::setJetThrusts) use the EKF estimate of that sub-step; the NN's own output is kept only as
 * its feedback state.  State: VSMPC_PS_TNN / _EST / _EKFP of the plant state (set them in the state handed to
 * vsmpc_rollout_reset).  Q, R: 2x2 row-major (ironcub_mujoco_simulator.py:54-56).  j == NULL switches back to the
 * polynomial jet plant.  The jet handle must outlive the rollout and live on the same device. */
struct vsmpc_rollout;
int vsmpc_rollout_set_jet_plant(struct vsmpc_rollout* r, vsmpc_jet* j, const double* Q, const double* R);

#ifdef __cplusplus
}
#endif
#endif /* VSMPC_JET_H */
