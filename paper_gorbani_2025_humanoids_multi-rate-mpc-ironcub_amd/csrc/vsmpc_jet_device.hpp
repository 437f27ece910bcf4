// Device code shared by the jet kernels (vsmpc_jet.hip) and the rollout's jet plant option (vsmpc_rollout.hip):
// one LSTM step from the zero state, input normalisation, the 2-state EKF update.
//   src/mujoco_lib/nn_jet_model.py:3-30,64-109, src/mujoco_lib/jet_kalman_filter.py:29-66 (paths relative to /root/reference/)
#pragma once
#include "vsmpc_device.hpp"

namespace vsmpc {

constexpr int JET_HMAX = 128;

struct JetNorm {
    double thrust_mean, thrust_std, throttle_mean, throttle_std;
};

VS_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// weights in LDS: wih0[4H] | wih1[4H] | b[4H] (= b_ih + b_hh, summed in the order torch adds them) | fc_w[H] | fc_b
template <bool WITH_STATE>
VS_DEV void lstm_step_from_zero(const float* __restrict__ sw, int H, float x0, float x1, float& out, float* h_out,
                                float* c_out, size_t stride) {
    const float* wih0 = sw;
    const float* wih1 = sw + 4 * H;
    const float* bih = sw + 8 * H;
    const float* bhh = sw + 12 * H;
    const float* fcw = sw + 16 * H;
    float acc = 0.0f;
    for (int j = 0; j < H; ++j) {
        // gates = W_ih x + b_ih + (W_hh 0) + b_hh   (torch adds the input and the hidden projections)
        const float gi = (wih0[j] * x0 + wih1[j] * x1 + bih[j]) + bhh[j];
        const float gg = (wih0[2 * H + j] * x0 + wih1[2 * H + j] * x1 + bih[2 * H + j]) + bhh[2 * H + j];
        const float go = (wih0[3 * H + j] * x0 + wih1[3 * H + j] * x1 + bih[3 * H + j]) + bhh[3 * H + j];
        const float c = sigmoidf_(gi) * tanhf(gg);       // f * c_prev = 0
        const float h = sigmoidf_(go) * tanhf(c);
        acc += fcw[j] * h;
        if (WITH_STATE) {
            if (h_out) h_out[j * stride] = h;
            if (c_out) c_out[j * stride] = c;
        }
    }
    out = acc + sw[17 * H];
}

VS_DEV void stage_weights(const float* __restrict__ gw, float* __restrict__ sw, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sw[i] = gw[i];
    __syncthreads();
}

// thrust [N], throttle [percent] -> normalised float32 inputs (nn_jet_model.py:64-73: Python-float arithmetic, then a
// float32 tensor)
VS_DEV void jet_normalize(const JetNorm& nm, float thrust, float throttle, float& x0, float& x1) {
    x0 = float((double(thrust) - nm.thrust_mean) / nm.thrust_std);
    x1 = float((double(throttle) - nm.throttle_mean) / nm.throttle_std);
}

// ---- second-order polynomial jet model + EKF (jet_kalman_filter.py:29-66), float64 ------------------------------------
struct Ekf2 {
    double q[4], r[4];
};

VS_DEV void ekf_update_dev(double& T, double& Td, double (&P)[4], double u, double zT, double zTd, double dt, const Ekf2& cv) {
    // x = f(x, u): T_dot first, T with the new T_dot (:36-44)
    const double a = (T - Jet::muT) / Jet::sgT, b = Td / Jet::sgT, us = (u - Jet::muU) / Jet::sgU;
    const double v = us + Jet::c12 * us * us;
    const double Tdd = Jet::f(a, b) + Jet::g(a, b) * v;
    Td = Td + Tdd * Jet::sgT * dt;
    T = T + Td * dt;
    // A = df/dx at the PREDICTED state (:59: self.A(x, u) after x = self.f(x, u))
    const double a2 = (T - Jet::muT) / Jet::sgT, b2 = Td / Jet::sgT;
    const double h_a = Jet::df_dT(a2, b2) + Jet::dg_dT(a2, b2) * v, h_b = Jet::df_dTd(a2, b2) + Jet::dg_dTd(a2, b2) * v;
    const double A10 = dt * h_a, A11 = 1.0 + dt * h_b;
    const double A00 = 1.0 + dt * A10, A01 = dt * A11;
    // P = A P A^T + Q
    const double AP00 = A00 * P[0] + A01 * P[2], AP01 = A00 * P[1] + A01 * P[3];
    const double AP10 = A10 * P[0] + A11 * P[2], AP11 = A10 * P[1] + A11 * P[3];
    double P00 = AP00 * A00 + AP01 * A01 + cv.q[0], P01 = AP00 * A10 + AP01 * A11 + cv.q[1];
    double P10 = AP10 * A00 + AP11 * A01 + cv.q[2], P11 = AP10 * A10 + AP11 * A11 + cv.q[3];
    // S = P + R, K = P S^-1 (H = I)
    const double S00 = P00 + cv.r[0], S01 = P01 + cv.r[1], S10 = P10 + cv.r[2], S11 = P11 + cv.r[3];
    const double idet = 1.0 / (S00 * S11 - S01 * S10);
    const double Si00 = S11 * idet, Si01 = -S01 * idet, Si10 = -S10 * idet, Si11 = S00 * idet;
    const double K00 = P00 * Si00 + P01 * Si10, K01 = P00 * Si01 + P01 * Si11;
    const double K10 = P10 * Si00 + P11 * Si10, K11 = P10 * Si01 + P11 * Si11;
    const double e0 = zT - T, e1 = zTd - Td;
    T += K00 * e0 + K01 * e1;
    Td += K10 * e0 + K11 * e1;
    // P = (I - K) P
    P[0] = (1.0 - K00) * P00 - K01 * P10;
    P[1] = (1.0 - K00) * P01 - K01 * P11;
    P[2] = -K10 * P00 + (1.0 - K11) * P10;
    P[3] = -K10 * P01 + (1.0 - K11) * P11;
}

// One LSTM step from the zero state with the hidden units spread over the 64 lanes of a wavefront (unit = lane, lane +
// 64, ...): every lane returns the network output fc(h) (normalised thrust rate).  Same arithmetic per unit as
// lstm_step_from_zero; the sum over the units is a butterfly instead of a serial loop (float32: last-bit differences).
VS_DEV float lstm_step_from_zero_wave(const float* __restrict__ sw, int H, float x0, float x1, int lane) {
    const float* wih0 = sw;
    const float* wih1 = sw + 4 * H;
    const float* bih = sw + 8 * H;
    const float* bhh = sw + 12 * H;
    const float* fcw = sw + 16 * H;
    float acc = 0.0f;
    for (int j = lane; j < H; j += 64) {
        const float gi = (wih0[j] * x0 + wih1[j] * x1 + bih[j]) + bhh[j];
        const float gg = (wih0[2 * H + j] * x0 + wih1[2 * H + j] * x1 + bih[2 * H + j]) + bhh[2 * H + j];
        const float go = (wih0[3 * H + j] * x0 + wih1[3 * H + j] * x1 + bih[3 * H + j]) + bhh[3 * H + j];
        const float c = sigmoidf_(gi) * tanhf(gg);
        acc += fcw[j] * (sigmoidf_(go) * tanhf(c));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc + sw[17 * H];
}

}  // namespace vsmpc
