// Test driver for include/VariableSamplingMPC.hpp: replays input records through the reference-shaped C++ class.
//   host_wrapper_driver <records.bin> <n_records> <out.bin> [apply_tick_state]
// out per record: status(1) | first-move-equivalent getters: dq via q_ref delta (8) | throttle(4) | thrust(4) |
// thrustDot(4) | jointsReference(23) | finalCoM(3) | hold flag used (1)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "VariableSamplingMPC.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int n = std::atoi(argv[2]);
    const bool tick = argc > 4 && std::atoi(argv[4]) != 0;
    vsmpc_config c{};
    c.n_iter = 17; c.n_iter_small = 7; c.control_horizon = 12; c.use_jet_dynamic = 1;
    c.period_mpc = 0.005; c.period_small = 0.005; c.period_large = 0.1;
    const double w1[3] = {500, 500, 5000}, w2[3] = {25000, 25000, 50000}, w3[3] = {1, 1, 1.5}, w4[3] = {1000, 1000, 1000},
                 w5[3] = {10000, 10000, 10000}, w6[3] = {80, 80, 80};
    for (int i = 0; i < 3; ++i) { c.w_com_pos[i] = w1[i]; c.w_com_pos_err[i] = w2[i]; c.w_lin_mom[i] = w3[i]; c.w_rpy[i] = w4[i]; c.w_rpy_err[i] = w5[i]; c.w_ang_mom[i] = w6[i]; }
    for (int i = 0; i < 8; ++i) c.w_delta_joint[i] = 65000.0;
    c.w_throttle = 80000.0; c.w_initial_throttle = 80000.0; c.w_reg_joint_pos = 20.0; c.throttle_min = 0.0; c.throttle_max = 100.0;

    vsmpc_host::VariableSamplingMPC mpc;
    std::vector<double> q0(23, 0.0);
    const double rpy0[3] = {0, 0, 0};
    if (!mpc.configure(c, q0.data(), rpy0, 0)) { std::fprintf(stderr, "configure failed: %s\n", vsmpc_strerror(mpc.getLastError())); return 3; }
    const int nin = 294;
    std::vector<double> recs(size_t(n) * nin);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(recs.data(), sizeof(double), recs.size(), f) != recs.size()) return 4;
    std::fclose(f);
    FILE* o = std::fopen(argv[3], "wb");
    if (!o) return 5;
    if (mpc.getNStatesMPC() != 26.0 || mpc.getNInputMPC() != 12.0 || mpc.getNOptimizationVariables() != 588u) return 6;
    double bad3[2];
    if (mpc.getThrustReference(bad3, 2)) return 7;   // wrong size must be refused like the reference does
    for (int k = 0; k < n; ++k) {
        if (!mpc.update(&recs[size_t(k) * nin], tick)) return 8;
        if (!mpc.solveMPC()) return 9;
        double row[48] = {0};
        row[0] = mpc.getQPProblemStatus();
        double thr[4], T[4], Td[4], q[23], com[3];
        mpc.getThrottleReference(thr, 4); mpc.getThrustReference(T, 4); mpc.getThrustDotReference(Td, 4);
        mpc.getJointsReferencePosition(q, 23); mpc.getFinalCoMPosition(com, 3);
        for (int i = 0; i < 4; ++i) { row[1 + i] = thr[i]; row[5 + i] = T[i]; row[9 + i] = Td[i]; }
        for (int i = 0; i < 23; ++i) row[13 + i] = q[i];
        for (int i = 0; i < 3; ++i) row[36 + i] = com[i];
        std::fwrite(row, sizeof(double), 48, o);
    }
    std::fclose(o);
    return 0;
}
