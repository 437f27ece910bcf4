// VariableSamplingMPC.hpp — host-only C++ mirror of the reference class on top of the C-ABI (vsmpc.h).
//
// Same method names, return conventions and tick semantics as
//   momentum-based-linear-mpc-lib/include/variableSamplingMPC/variableSamplingMPC.h:15-41
//   momentum-based-linear-mpc-lib/include/IMPCProblem/IMPCProblem.h:35-148
// (paths relative to /root/reference/src/flight-controller/), so that a pybind shim with the reference's Python
// names (bindings/python/MPCPyBindings.cpp:22-90) is mechanical.  What differs, by construction of the boundary
// (SURVEY.md 8b): update() receives the already extracted per-tick record (VSMPC_IN_* layout) instead of a live
// QPInput/Robot, and `TickState` carries the per-instance state the reference hides inside its plugins:
//   * 20-tick throttle hold counter        constraintsVSMPC.cpp:335,351-372
//   * RPY unwrapping with turn counters    constraintsVSMPC.cpp:232-247
//   * joint-position accumulator           variableSamplingMPC.cpp:59-60,104-108
//   * "consume the solution only if Solved" variableSamplingMPC.cpp:91
// All numerics run in libvsmpc.so (HIP); there is no CPU solve path here.
#ifndef VARIABLE_SAMPLING_MPC_HPP
#define VARIABLE_SAMPLING_MPC_HPP

#include <cmath>
#include <vector>

#include "vsmpc.h"

namespace vsmpc_host {

constexpr int kRobotJoints = 23;   // MPCPyBindings.cpp:43 hard-codes 23 joints
constexpr int kJointOffset = 3;    // controlled joints are robot joints 3..10 (systemDynamicsVSMPC.cpp:348)
constexpr double kPi = 3.14159265358979323846;

// Per-instance tick state machine.
struct TickState {
    int ratio = 20;          // round(periodMPCLargeSteps / periodMPCSmallSteps), constraintsVSMPC.cpp:322
    int throttleCounter = 0; // ThrottleConstraint::m_counter
    double rpyOld[3] = {0, 0, 0};
    double nTurns[3] = {0, 0, 0};

    void configure(const vsmpc_config& c, const double rpy0[3]) {
        ratio = static_cast<int>(std::lround(c.period_large / c.period_small));
        // m_counter starts at ratio-1 (constraintsVSMPC.cpp:335) and IMPCProblem::configure evaluates the
        // constraint once (IMPCProblem.cpp:94-96), which wraps it to 0.
        throttleCounter = 0;
        for (int i = 0; i < 3; ++i) { rpyOld[i] = rpy0[i]; nTurns[i] = 0.0; }
    }
    // hold flag of THIS tick, then advance (constraintsVSMPC.cpp:351,366-372)
    bool nextHoldFlag() {
        const bool hold = throttleCounter != ratio - 1;
        throttleCounter = (throttleCounter == ratio - 1) ? 0 : throttleCounter + 1;
        return hold;
    }
    // constraintsVSMPC.cpp:232-247
    void unwrapRPY(const double rpy[3], double out[3]) {
        for (int i = 0; i < 3; ++i) {
            if (rpy[i] - rpyOld[i] > kPi) nTurns[i] -= 1.0;
            else if (rpy[i] - rpyOld[i] < -kPi) nTurns[i] += 1.0;
            out[i] = rpy[i] + 2.0 * kPi * nTurns[i];
            rpyOld[i] = rpy[i];
        }
    }
};

class VariableSamplingMPC {
public:
    VariableSamplingMPC() = default;
    ~VariableSamplingMPC() { if (m_h) vsmpc_destroy(m_h); }
    VariableSamplingMPC(const VariableSamplingMPC&) = delete;
    VariableSamplingMPC& operator=(const VariableSamplingMPC&) = delete;

    // IMPCProblem::configure (IMPCProblem.cpp:3-148): sizes, buffers, kernel selection.
    bool configure(const vsmpc_config& cfg, const double* initialJointPositions /* 23 or nullptr */,
                   const double initialRPY[3], int device = 0) {
        if (m_h) { vsmpc_destroy(m_h); m_h = nullptr; }
        m_lastError = vsmpc_create(&cfg, device, 1, &m_h);
        if (m_lastError != VSMPC_OK) return false;
        m_cfg = cfg;
        m_nVar = vsmpc_num_variables(m_h);
        m_nIn = vsmpc_input_doubles(m_h);
        m_nStates = VSMPC_N_STATES;                                  // variableSamplingMPC.cpp:42
        m_nInput = VSMPC_N_JOINTS + VSMPC_N_THRUSTS;                 // :43
        m_record.assign(m_nIn, 0.0);
        m_QPSolution.assign(m_nVar, 0.0);
        m_firstMove.assign(VSMPC_FM_SIZE, 0.0);
        m_jointsPositionReference.assign(kRobotJoints, 0.0);         // :59-60
        if (initialJointPositions)
            for (int i = 0; i < kRobotJoints; ++i) m_jointsPositionReference[i] = initialJointPositions[i];
        m_deltaJoints.assign(VSMPC_N_JOINTS, 0.0);
        m_thrust.assign(4, 0.0); m_thrustDot.assign(4, 0.0); m_throttle.assign(4, 0.0);
        m_finalState.assign(VSMPC_N_STATES, 0.0);
        const double zero[3] = {0, 0, 0};
        m_tick.configure(cfg, initialRPY ? initialRPY : zero);
        m_status = 0;
        m_haveRecord = false;
        return true;
    }

    // IMPCProblem::update (IMPCProblem.cpp:150-194).  `record` is the VSMPC_IN_* image of this tick.  When
    // `applyTickState` is true the wrapper overwrites the hold flag and the unwrapped RPY entries of X0 from its
    // own state machine (as the reference's plugins would); otherwise the record is used verbatim.
    bool update(const double* record, bool applyTickState = false) {
        if (!m_h || !record) return false;
        for (int i = 0; i < m_nIn; ++i) m_record[i] = record[i];
        if (applyTickState) {
            m_record[VSMPC_IN_HOLD] = m_tick.nextHoldFlag() ? 1.0 : 0.0;
            double un[3];
            m_tick.unwrapRPY(&m_record[VSMPC_IN_RPY], un);
            for (int i = 0; i < 3; ++i) {
                const double ref = m_record[VSMPC_IN_X0 + 6 + i] - m_record[VSMPC_IN_X0 + 23 + i];  // RPYReference
                m_record[VSMPC_IN_X0 + 6 + i] = un[i];                                              // :212
                m_record[VSMPC_IN_X0 + 23 + i] = un[i] - ref;                                       // :227-228
            }
        }
        m_haveRecord = true;
        return true;
    }

    // VariableSamplingMPC::solveMPC (variableSamplingMPC.cpp:88-112): returns true regardless, consumes the
    // solution only if the status is Solved.
    bool solveMPC() {
        if (!m_h || !m_haveRecord) return true;
        std::vector<double> x(m_nVar), fm(VSMPC_FM_SIZE);
        int status = 0;
        m_lastError = vsmpc_solve_batch(m_h, m_record.data(), 1, x.data(), fm.data(), &status, nullptr, nullptr);
        m_status = (m_lastError == VSMPC_OK) ? status : VSMPC_STATUS_NUMERICAL;
        if (m_status == VSMPC_STATUS_SOLVED) {
            m_QPSolution = x;
            m_firstMove = fm;
            for (int i = 0; i < VSMPC_N_JOINTS; ++i) m_deltaJoints[i] = fm[VSMPC_FM_DQ + i];
            for (int i = 0; i < 4; ++i) {
                m_throttle[i] = fm[VSMPC_FM_THROTTLE + i];
                m_thrust[i] = fm[VSMPC_FM_THRUST + i];
                m_thrustDot[i] = fm[VSMPC_FM_THRUSTDOT + i];
            }
            const int N = m_cfg.n_iter;
            for (int i = 0; i < VSMPC_N_STATES; ++i) m_finalState[i] = x[VSMPC_N_STATES * N + i];
            for (int i = 0; i < VSMPC_N_JOINTS; ++i)
                m_jointsPositionReference[kJointOffset + i] += m_deltaJoints[i];                    // :104-108
        }
        return true;
    }

    int getQPProblemStatus() const { return m_status; }
    int getLastError() const { return m_lastError; }
    unsigned int getNOptimizationVariables() const { return static_cast<unsigned int>(m_nVar); }

    // getters: size-checked like the reference (variableSamplingMPC.cpp:114-217), returning bool
    bool getMPCSolution(double* out, int size) const {
        const int n = m_nVar - VSMPC_N_STATES * (m_cfg.n_iter + 1);
        if (size != n) return false;
        for (int i = 0; i < n; ++i) out[i] = m_QPSolution[VSMPC_N_STATES * (m_cfg.n_iter + 1) + i];
        return true;
    }
    bool getJointsReferencePosition(double* out, int size) const { return copy(m_jointsPositionReference, out, size); }
    bool getThrottleReference(double* out, int size) const { return copy(m_throttle, out, size); }
    bool getThrustReference(double* out, int size) const { return copy(m_thrust, out, size); }
    bool getThrustDotReference(double* out, int size) const { return copy(m_thrustDot, out, size); }
    bool getFinalCoMPosition(double* out, int size) const { return slice(m_finalState, 0, out, size); }
    bool getFinalLinMom(double* out, int size) const { return slice(m_finalState, 3, out, size); }
    bool getFinalRPY(double* out, int size) const { return slice(m_finalState, 6, out, size); }
    bool getFinalAngMom(double* out, int size) const { return slice(m_finalState, 9, out, size); }
    double getNStatesMPC() const { return m_nStates; }   // the reference returns double (variableSamplingMPC.cpp:219-227)
    double getNInputMPC() const { return m_nInput; }
    const std::vector<double>& getSolution() const { return m_QPSolution; }   // IMPCProblem::getSolution
    TickState& tickState() { return m_tick; }
    const vsmpc_config& config() const { return m_cfg; }
    int inputDoubles() const { return m_nIn; }

private:
    static bool copy(const std::vector<double>& v, double* out, int size) {
        if (size != static_cast<int>(v.size())) return false;
        for (int i = 0; i < size; ++i) out[i] = v[i];
        return true;
    }
    static bool slice(const std::vector<double>& v, int off, double* out, int size) {
        if (size != 3) return false;
        for (int i = 0; i < 3; ++i) out[i] = v[off + i];
        return true;
    }
    vsmpc_handle* m_h = nullptr;
    vsmpc_config m_cfg{};
    TickState m_tick;
    int m_nVar = 0, m_nIn = 0, m_status = 0, m_lastError = 0;
    int m_nStates = 0, m_nInput = 0;
    bool m_haveRecord = false;
    std::vector<double> m_record, m_QPSolution, m_firstMove, m_jointsPositionReference, m_deltaJoints;
    std::vector<double> m_thrust, m_thrustDot, m_throttle, m_finalState;
};

}  // namespace vsmpc_host
#endif  // VARIABLE_SAMPLING_MPC_HPP
