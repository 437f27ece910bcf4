// Batched jet plant / estimator kernels and their C-ABI (include/vsmpc_jet.h), gfx950.
//
//   jet_nn_step_kernel      JetModelTotal.get_state (src/mujoco_lib/nn_jet_model.py:86-109): per series (one jet of one
//                           instance) ONE LSTM(2 -> H) step from zero state + Linear(H -> 1), float32 like the reference
//   jet_nn_sequence_kernel  NeuralJetModel.get_state (:16-30) on sequences, state carried (exercises W_hh)
//   jet_ekf_kernel          SecondOrderJetModel.update (src/mujoco_lib/jet_kalman_filter.py:57-66), float64
//   jet_plant_kernel        MujocoSim.step with use_nn_jet_dynamics (ironcub_mujoco_simulator.py:128-133,393-396):
//                           `steps` x (NN step with the thrust fed back -> EKF update), all state in registers
//
// Roofline: every kernel here is latency / transcendental bound on a tiny working set (weights 2.6 KB for the one-step
// path, 102 KB with W_hh), a few KB of HBM traffic per thousand series: nothing to tile for the matrix cores -- the
// reference calls the LSTM with sequence length 1 and zero state, so W_hh h is identically zero on its path and the
// "GEMV" is two multiply-adds per gate.  One series per lane, weights broadcast from LDS.
#include <cmath>
#include <cstdio>
#include <new>

#include "vsmpc_device.hpp"
#include "vsmpc_jet_device.hpp"
#include "vsmpc_launch.hpp"
#include "../../include/vsmpc_jet.h"

namespace vsmpc {

__global__ __launch_bounds__(256) void jet_nn_step_kernel(const float* __restrict__ w, int H, JetNorm nm,
                                                          const float* __restrict__ thrust, const float* __restrict__ throttle,
                                                          int n, float dt, float* __restrict__ T_next, float* __restrict__ T_dot,
                                                          float* __restrict__ h_out, float* __restrict__ c_out) {
    __shared__ float sw[17 * JET_HMAX + 1];
    stage_weights(w, sw, 17 * H + 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x0, x1, out;
    jet_normalize(nm, thrust[i], throttle[i], x0, x1);
    if (h_out != nullptr || c_out != nullptr)
        lstm_step_from_zero<true>(sw, H, x0, x1, out, h_out ? h_out + size_t(i) * H : nullptr,
                                  c_out ? c_out + size_t(i) * H : nullptr, 1);
    else
        lstm_step_from_zero<false>(sw, H, x0, x1, out, nullptr, nullptr, 1);
    const float tn = x0 + out * dt;                                          // nn_jet_model.py:27
    T_next[i] = tn * float(nm.thrust_std) + float(nm.thrust_mean);          // _denormalize_thrust (:75-77)
    T_dot[i] = out * float(nm.thrust_std);                                   // _denormalize_thrust_dot (:79-81)
}

// One workgroup of H threads (rounded up to a wavefront multiple) per sequence: thread j owns hidden unit j, the hidden
// vector lives in LDS, W_hh^T ([H][4H], transposed at create time) is read coalesced from L2.
__global__ __launch_bounds__(JET_HMAX) void jet_nn_sequence_kernel(const float* __restrict__ w, const float* __restrict__ whhT,
                                                                   int H, const float* __restrict__ x, int n, int L, float dt,
                                                                   float* __restrict__ T_next_norm, float* __restrict__ T_dot_norm,
                                                                   float* __restrict__ h_n, float* __restrict__ c_n) {
    __shared__ float sw[17 * JET_HMAX + 1];
    __shared__ float sh[JET_HMAX], sred[JET_HMAX];
    stage_weights(w, sw, 17 * H + 1);
    const int s = blockIdx.x, j = threadIdx.x;
    if (s >= n) return;
    const bool act = j < H;
    float c = 0.0f, h = 0.0f;
    if (act) sh[j] = 0.0f;
    __syncthreads();
    const float* xs = x + size_t(s) * L * 2;
    for (int t = 0; t < L; ++t) {
        const float x0 = xs[2 * t], x1 = xs[2 * t + 1];
        float g[4];
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * H + j;
                float hh = 0.0f;
                for (int k = 0; k < H; ++k) hh += whhT[size_t(k) * 4 * H + r] * sh[k];
                g[q] = (sw[r] * x0 + sw[4 * H + r] * x1 + sw[8 * H + r]) + (hh + sw[12 * H + r]);
            }
        }
        __syncthreads();                    // everyone has read the old hidden vector
        if (act) {
            c = sigmoidf_(g[1]) * c + sigmoidf_(g[0]) * tanhf(g[2]);
            h = sigmoidf_(g[3]) * tanhf(c);
            sh[j] = h;
        }
        __syncthreads();
    }
    sred[j] = act ? sw[16 * H + j] * h : 0.0f;
    __syncthreads();
    if (j == 0) {
        float acc = 0.0f;
        for (int k = 0; k < H; ++k) acc += sred[k];
        const float out = acc + sw[17 * H];
        T_dot_norm[s] = out;
        T_next_norm[s] = xs[2 * (L - 1)] + out * dt;
    }
    if (act) {
        if (h_n) h_n[size_t(s) * H + j] = h;
        if (c_n) c_n[size_t(s) * H + j] = c;
    }
}

__global__ __launch_bounds__(256) void jet_ekf_kernel(double* __restrict__ x, double* __restrict__ P, const double* __restrict__ u,
                                                      const double* __restrict__ z, int n, double dt, Ekf2 cv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double T = x[2 * i], Td = x[2 * i + 1];
    double Pm[4] = {P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]};
    ekf_update_dev(T, Td, Pm, u[i], z[2 * i], z[2 * i + 1], dt, cv);
    x[2 * i] = T;
    x[2 * i + 1] = Td;
    for (int k = 0; k < 4; ++k) P[4 * i + k] = Pm[k];
}

__global__ __launch_bounds__(256) void jet_plant_kernel(const float* __restrict__ w, int H, JetNorm nm, float* __restrict__ T_nn,
                                                        double* __restrict__ x_est, double* __restrict__ P,
                                                        const float* __restrict__ throttle, int throttle_steps, int n, int steps,
                                                        double dt, Ekf2 cv, double* __restrict__ log) {
    __shared__ float sw[17 * JET_HMAX + 1];
    stage_weights(w, sw, 17 * H + 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float Tn = T_nn[i];
    double T = x_est[2 * i], Td = x_est[2 * i + 1];
    double Pm[4] = {P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]};
    const float dtf = float(dt);
    for (int k = 0; k < steps; ++k) {
        const float u = throttle[(throttle_steps == 1 ? 0 : size_t(k) * n) + i];
        float x0, x1, out;
        jet_normalize(nm, Tn, u, x0, x1);
        lstm_step_from_zero<false>(sw, H, x0, x1, out, nullptr, nullptr, 1);
        Tn = (x0 + out * dtf) * float(nm.thrust_std) + float(nm.thrust_mean);   // _simulate_thrust_nn_model: T fed back
        const float Tdn = out * float(nm.thrust_std);
        ekf_update_dev(T, Td, Pm, double(u), double(Tn), double(Tdn), dt, cv);   // jet_EKF.update(..., T_nn, Tdot_nn)
        if (log != nullptr) {
            log[(size_t(k) * n + i) * 2] = T;
            log[(size_t(k) * n + i) * 2 + 1] = Td;
        }
    }
    T_nn[i] = Tn;
    x_est[2 * i] = T;
    x_est[2 * i + 1] = Td;
    for (int k = 0; k < 4; ++k) P[4 * i + k] = Pm[k];
}

}  // namespace vsmpc

using namespace vsmpc;

struct vsmpc_jet {
    int device, hidden, max_series;
    JetNorm nm;
    float* d_w;      // wih col 0 [4H] | wih col 1 [4H] | b_ih [4H] | b_hh [4H] | fc_w [H] | fc_b
    float* d_whhT;   // [H][4H]
    float *d_f0, *d_f1, *d_f2, *d_f3;   // float staging, max_series each
    float *d_h, *d_c;                    // [max_series][H]
    double *d_x, *d_P, *d_u, *d_z;       // EKF staging
    double* d_log;
    size_t log_doubles;
    float* d_thr_steps;
    size_t thr_floats;
};

namespace {
thread_local char g_jet_msg[256] = "";
int jet_fail(hipError_t e, const char* what) {
    snprintf(g_jet_msg, sizeof(g_jet_msg), "HIP error in %s: %s", what, hipGetErrorString(e));
    return VSMPC_ERR_HIP;
}
#define JET_TRY(expr)                                     \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return jet_fail(_e, #expr); \
    } while (0)

Ekf2 make_cov(const double* Q, const double* R) {
    Ekf2 cv;
    for (int k = 0; k < 4; ++k) { cv.q[k] = Q[k]; cv.r[k] = R[k]; }
    return cv;
}
}  // namespace

namespace vsmpc {
void jet_plant_view(const ::vsmpc_jet* j, const float** w, int* hidden, double norm[4], int* device) {
    *w = j->d_w;
    *hidden = j->hidden;
    *device = j->device;
    norm[0] = j->nm.thrust_mean; norm[1] = j->nm.thrust_std; norm[2] = j->nm.throttle_mean; norm[3] = j->nm.throttle_std;
}
}  // namespace vsmpc

extern "C" {

const char* vsmpc_jet_last_error(void) { return g_jet_msg; }

int vsmpc_jet_create(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* fc_w,
                     const float* fc_b, const double* norm, int hidden, int device, int max_series, vsmpc_jet** out) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !fc_w || !fc_b || !norm || !out || hidden <= 0 || hidden > JET_HMAX ||
        max_series <= 0 || !(norm[1] > 0.0) || !(norm[3] > 0.0))
        return VSMPC_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    JET_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return VSMPC_ERR_INVALID_ARG;
    vsmpc::DeviceScope _scope(device); JET_TRY(_scope.err);
    vsmpc_jet* j = new (std::nothrow) vsmpc_jet();
    if (!j) return VSMPC_ERR_ALLOC;
    *j = vsmpc_jet{};
    j->device = device;
    j->hidden = hidden;
    j->max_series = max_series;
    j->nm = JetNorm{norm[0], norm[1], norm[2], norm[3]};
    const int H = hidden;
    float* hw = new (std::nothrow) float[17 * H + 1 + size_t(H) * 4 * H];
    if (!hw) { delete j; return VSMPC_ERR_ALLOC; }
    for (int r = 0; r < 4 * H; ++r) {
        hw[r] = w_ih[2 * r];
        hw[4 * H + r] = w_ih[2 * r + 1];
        hw[8 * H + r] = b_ih[r];
        hw[12 * H + r] = b_hh[r];
    }
    for (int k = 0; k < H; ++k) hw[16 * H + k] = fc_w[k];
    hw[17 * H] = fc_b[0];
    float* hT = hw + 17 * H + 1;
    for (int r = 0; r < 4 * H; ++r)
        for (int k = 0; k < H; ++k) hT[size_t(k) * 4 * H + r] = w_hh[size_t(r) * H + k];
    const size_t S = size_t(max_series);
    hipError_t e = hipMalloc(&j->d_w, (17 * H + 1) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_whhT, size_t(H) * 4 * H * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(j->d_w, hw, (17 * H + 1) * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(j->d_whhT, hT, size_t(H) * 4 * H * sizeof(float), hipMemcpyHostToDevice);
    delete[] hw;
    if (e == hipSuccess) e = hipMalloc(&j->d_f0, S * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_f1, S * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_f2, S * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_f3, S * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_h, S * H * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_c, S * H * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&j->d_x, S * 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&j->d_P, S * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&j->d_u, S * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&j->d_z, S * 2 * sizeof(double));
    if (e != hipSuccess) {
        vsmpc_jet_destroy(j);
        return e == hipErrorOutOfMemory ? VSMPC_ERR_ALLOC : jet_fail(e, "vsmpc_jet_create");
    }
    *out = j;
    return VSMPC_OK;
}

void vsmpc_jet_destroy(vsmpc_jet* j) {
    if (!j) return;
    vsmpc::DeviceScope _scope(j->device);
    void* ptrs[] = {j->d_w, j->d_whhT, j->d_f0, j->d_f1, j->d_f2, j->d_f3, j->d_h, j->d_c, j->d_x, j->d_P, j->d_u, j->d_z,
                    j->d_log, j->d_thr_steps};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete j;
}

int vsmpc_jet_nn_step(vsmpc_jet* j, const float* thrust, const float* throttle, int n, float dt, float* T_next,
                      float* T_dot, float* h_out, float* c_out) {
    if (!j || !thrust || !throttle || !T_next || !T_dot || n < 0) return VSMPC_ERR_INVALID_ARG;
    if (n > j->max_series) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (n == 0) return VSMPC_OK;
    vsmpc::DeviceScope _scope(j->device); JET_TRY(_scope.err);
    const size_t N = size_t(n);
    JET_TRY(hipMemcpy(j->d_f0, thrust, N * sizeof(float), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_f1, throttle, N * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(jet_nn_step_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, j->d_w, j->hidden, j->nm, j->d_f0,
                       j->d_f1, n, dt, j->d_f2, j->d_f3, h_out ? j->d_h : nullptr, c_out ? j->d_c : nullptr);
    JET_TRY(hipGetLastError());
    JET_TRY(hipDeviceSynchronize());
    JET_TRY(hipMemcpy(T_next, j->d_f2, N * sizeof(float), hipMemcpyDeviceToHost));
    JET_TRY(hipMemcpy(T_dot, j->d_f3, N * sizeof(float), hipMemcpyDeviceToHost));
    if (h_out) JET_TRY(hipMemcpy(h_out, j->d_h, N * j->hidden * sizeof(float), hipMemcpyDeviceToHost));
    if (c_out) JET_TRY(hipMemcpy(c_out, j->d_c, N * j->hidden * sizeof(float), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_jet_nn_sequence(vsmpc_jet* j, const float* x, int n, int L, float dt, float* T_next_norm, float* T_dot_norm,
                          float* h_n, float* c_n) {
    if (!j || !x || !T_next_norm || !T_dot_norm || n < 0 || L <= 0) return VSMPC_ERR_INVALID_ARG;
    if (n > j->max_series) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (n == 0) return VSMPC_OK;
    vsmpc::DeviceScope _scope(j->device); JET_TRY(_scope.err);
    float* d_x = nullptr;                       // sequences are a parity / offline entry: sized per call
    JET_TRY(hipMalloc(&d_x, size_t(n) * L * 2 * sizeof(float)));
    hipError_t e = hipMemcpy(d_x, x, size_t(n) * L * 2 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int threads = ((j->hidden + 63) / 64) * 64;
        hipLaunchKernelGGL(jet_nn_sequence_kernel, dim3(n), dim3(threads), 0, nullptr, j->d_w, j->d_whhT, j->hidden, d_x, n, L,
                           dt, j->d_f2, j->d_f3, h_n ? j->d_h : nullptr, c_n ? j->d_c : nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_x);
    if (e != hipSuccess) return jet_fail(e, "vsmpc_jet_nn_sequence");
    const size_t N = size_t(n);
    JET_TRY(hipMemcpy(T_next_norm, j->d_f2, N * sizeof(float), hipMemcpyDeviceToHost));
    JET_TRY(hipMemcpy(T_dot_norm, j->d_f3, N * sizeof(float), hipMemcpyDeviceToHost));
    if (h_n) JET_TRY(hipMemcpy(h_n, j->d_h, N * j->hidden * sizeof(float), hipMemcpyDeviceToHost));
    if (c_n) JET_TRY(hipMemcpy(c_n, j->d_c, N * j->hidden * sizeof(float), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_jet_ekf_update(vsmpc_jet* j, double* x, double* P, const double* u, const double* z, int n, double dt,
                         const double* Q, const double* R) {
    if (!j || !x || !P || !u || !z || !Q || !R || n < 0 || !(dt > 0.0)) return VSMPC_ERR_INVALID_ARG;
    if (n > j->max_series) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (n == 0) return VSMPC_OK;
    vsmpc::DeviceScope _scope(j->device); JET_TRY(_scope.err);
    const size_t N = size_t(n);
    JET_TRY(hipMemcpy(j->d_x, x, N * 2 * sizeof(double), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_P, P, N * 4 * sizeof(double), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_u, u, N * sizeof(double), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_z, z, N * 2 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(jet_ekf_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, j->d_x, j->d_P, j->d_u, j->d_z, n, dt,
                       make_cov(Q, R));
    JET_TRY(hipGetLastError());
    JET_TRY(hipDeviceSynchronize());
    JET_TRY(hipMemcpy(x, j->d_x, N * 2 * sizeof(double), hipMemcpyDeviceToHost));
    JET_TRY(hipMemcpy(P, j->d_P, N * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_jet_plant_run_device(vsmpc_jet* j, float* d_T_nn, double* d_x_est, double* d_P, const float* d_throttle,
                               int throttle_steps, int n, int steps, double dt, const double* Q, const double* R,
                               double* d_log, void* stream) {
    if (!j || !d_T_nn || !d_x_est || !d_P || !d_throttle || !Q || !R || n < 0 || steps < 0 || !(dt > 0.0) ||
        (throttle_steps != 1 && throttle_steps != steps))
        return VSMPC_ERR_INVALID_ARG;
    if (n == 0 || steps == 0) return VSMPC_OK;
    vsmpc::DeviceScope _scope(j->device); JET_TRY(_scope.err);
    hipLaunchKernelGGL(jet_plant_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), j->d_w,
                       j->hidden, j->nm, d_T_nn, d_x_est, d_P, d_throttle, throttle_steps, n, steps, dt, make_cov(Q, R), d_log);
    JET_TRY(hipGetLastError());
    return VSMPC_OK;
}

int vsmpc_jet_plant_run(vsmpc_jet* j, float* T_nn, double* x_est, double* P, const float* throttle, int throttle_steps,
                        int n, int steps, double dt, const double* Q, const double* R, double* log) {
    if (!j || !T_nn || !x_est || !P || !throttle || !Q || !R || n < 0 || steps < 0 ||
        (throttle_steps != 1 && throttle_steps != steps))
        return VSMPC_ERR_INVALID_ARG;
    if (n > j->max_series) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (n == 0 || steps == 0) return VSMPC_OK;
    vsmpc::DeviceScope _scope(j->device); JET_TRY(_scope.err);
    const size_t N = size_t(n), TS = size_t(throttle_steps) * N;
    if (TS > j->thr_floats) {              // grows only when a longer schedule than ever before is passed
        if (j->d_thr_steps) (void)hipFree(j->d_thr_steps);
        j->d_thr_steps = nullptr;
        j->thr_floats = 0;
        JET_TRY(hipMalloc(&j->d_thr_steps, TS * sizeof(float)));
        j->thr_floats = TS;
    }
    const size_t LG = log ? size_t(steps) * N * 2 : 0;
    if (LG > j->log_doubles) {
        if (j->d_log) (void)hipFree(j->d_log);
        j->d_log = nullptr;
        j->log_doubles = 0;
        JET_TRY(hipMalloc(&j->d_log, LG * sizeof(double)));
        j->log_doubles = LG;
    }
    JET_TRY(hipMemcpy(j->d_f0, T_nn, N * sizeof(float), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_x, x_est, N * 2 * sizeof(double), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_P, P, N * 4 * sizeof(double), hipMemcpyHostToDevice));
    JET_TRY(hipMemcpy(j->d_thr_steps, throttle, TS * sizeof(float), hipMemcpyHostToDevice));
    int rc = vsmpc_jet_plant_run_device(j, j->d_f0, j->d_x, j->d_P, j->d_thr_steps, throttle_steps, n, steps, dt, Q, R,
                                        log ? j->d_log : nullptr, nullptr);
    if (rc != VSMPC_OK) return rc;
    JET_TRY(hipDeviceSynchronize());
    JET_TRY(hipMemcpy(T_nn, j->d_f0, N * sizeof(float), hipMemcpyDeviceToHost));
    JET_TRY(hipMemcpy(x_est, j->d_x, N * 2 * sizeof(double), hipMemcpyDeviceToHost));
    JET_TRY(hipMemcpy(P, j->d_P, N * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (log) JET_TRY(hipMemcpy(log, j->d_log, LG * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

}  // extern "C"
