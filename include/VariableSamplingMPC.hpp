// VariableSamplingMPC.hpp — host-only C++ mirror of the reference class on top of the C-ABI (vsmpc.h).
//
// One implementation of everything that sits between the reference's caller and the device path:
//
//   TickMachine                 the per-instance state the reference hides inside its plugins, and the PACKER that turns
//                               one tick of (Robot, Robot reference, QPInput) into the vsmpc_input record:
//                                 reference window FIFO + trajectory cursor      costsVSMPC.cpp:96-181
//                                 joint-posture error, name-based joint selection costsVSMPC.cpp:539-590
//                                 linearisation inputs, alpha-gravity cursor      systemDynamicsVSMPC.cpp:72-226,288-350,384-429
//                                 X0, RPY unwrap with turn counters               constraintsVSMPC.cpp:184-247
//                                 20-tick throttle hold                           constraintsVSMPC.cpp:335-372
//                                 solution slicing, joint accumulator, consume-only-if-Solved  variableSamplingMPC.cpp:88-151
//                               It works on plain snapshots (RobotView, QPView), so every front end shares it.
//   readRobot / readQPInput     templates that fill those snapshots through EXACTLY the getters utils/include/Robot.h and
//                               utils/include/QPInput.h:12-124 have (element access with operator()), i.e. they compile
//                               against the reference's own classes inside momentum-based-linear-mpc-lib.
//   VariableSamplingMPCT<...>   the reference's surface -- configure(parametersHandler, qpInput) / update(qpInput) /
//                               solveMPC() / get*Reference(out) (variableSamplingMPC.h:15-41, IMPCProblem.h:35-148,
//                               MPCPyBindings.cpp:22-90) -- over TickMachine + the C-ABI.  The class a maintainer adds
//                               to the reference is VariableSamplingMPCT<QPInput, TrajectoryManager> (INTEGRATION.md
//                               section 2).  What that instantiation meets inside the reference, and what this header
//                               therefore does NOT assume:
//                                 * QPInput's setters take Eigen::Ref<const Eigen::Vector3d> / const Eigen::Vector6d&
//                                   (QPInput.h:41,45,67): nothing converts to those from a double*.  Arguments are
//                                   built by fixedArg<N>() -- an Eigen::Map when <Eigen/Core> can be included, a small
//                                   view with operator()/data() otherwise;
//                                 * TrajectoryManager::getCurrentValue returns an Eigen::VectorXd BY VALUE
//                                   (TrajectoryManager.h:104) and there is no has(): values are kept alive in a local
//                                   and read with [], optional tracks are probed only where the type offers has();
//                                 * configure receives a std::weak_ptr<IParametersHandler> (IMPCProblem.h:35).
//                               tests/cpp/reference_surface_driver.cpp builds exactly this instantiation against
//                               signature-exact stand-ins of those headers (tests/cpp/refstub/).
//   VariableSamplingMPC         record-level front end (update(record)): the caller already holds the vsmpc_input record.
//
// All numerics run in libvsmpc.so (HIP): Lambda_lin / Lambda_ang / I_G through vsmpc_kinematics_batch, update()+solveMPC()
// through vsmpc_solve_batch.  There is no CPU solve path here.  Paths are relative to
// /root/reference/src/flight-controller/.
#ifndef VARIABLE_SAMPLING_MPC_HPP
#define VARIABLE_SAMPLING_MPC_HPP

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "vsmpc.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define VSMPC_HOST_HAS_EIGEN 1
#endif
#endif

namespace vsmpc_host {

// ---------------------------------------------------------------------------------------------------------------------
// fixedArg<N>(p): the argument handed to a QPInput setter for N doubles at p.  Inside the reference (Eigen present) an
// Eigen::Map<const Matrix<double, N, 1>>: converts implicitly to Eigen::Ref<const Vector3d> (QPInput.h:41,45) and to a
// const Vector6d& temporary (QPInput.h:67).  Without Eigen (this repo's own front ends) a view with the same read access.
// ---------------------------------------------------------------------------------------------------------------------
template <int N>
struct FixedView {
    const double* p;
    double operator()(int i) const { return p[i]; }
    double operator[](int i) const { return p[i]; }
    const double* data() const { return p; }
    int size() const { return N; }
};
#ifdef VSMPC_HOST_HAS_EIGEN
template <int N>
inline Eigen::Map<const Eigen::Matrix<double, N, 1>> fixedArg(const double* p) {
    return Eigen::Map<const Eigen::Matrix<double, N, 1>>(p);
}
#else
template <int N>
inline FixedView<N> fixedArg(const double* p) { return FixedView<N>{p}; }
#endif

// hasTrack(traj, name): TrajT::has(name) where the type has one (vsmpc_host::Trajectory); the reference's
// TrajectoryManager has none and loads every variable of its MAT file, RPY and RPYDot included (TrajectoryManager.cpp:
// 95-127; a missing key throws from map::at, :160-167) -- there every track the reference reads is taken as present.
template <class T, class = void>
struct HasTrackProbe : std::false_type {};
template <class T>
struct HasTrackProbe<T, std::void_t<decltype(std::declval<const T&>().has(std::declval<const std::string&>()))>> : std::true_type {};
template <class T>
inline bool hasTrack(const T& traj, const std::string& name) {
    if constexpr (HasTrackProbe<T>::value) return traj.has(name);
    else return true;
}

constexpr int kRobotJoints = VSMPC_KIN_NJ;   // MPCPyBindings.cpp:43 hard-codes 23 joints; the kinematics record does too
constexpr int kJets = VSMPC_N_THRUSTS;
constexpr int kJointOffset = 3;              // Lambda_lin's hard-coded column offset (systemDynamicsVSMPC.cpp:348)
constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------------------------------
// small dense helpers (row-major 3x3)
// ---------------------------------------------------------------------------------------------------------------------
inline void rpyOfRotation(const double R[9], double rpy[3]) {   // iDynTree::Rotation::asRPY, R = Rz(y) Ry(p) Rx(r)
    rpy[0] = std::atan2(R[7], R[8]);
    rpy[1] = std::atan2(-R[6], std::hypot(R[7], R[8]));
    rpy[2] = std::atan2(R[3], R[0]);
}
inline void mulRt(const double R[9], const double v[3], double out[3]) {   // R^T v
    for (int i = 0; i < 3; ++i) out[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}

// ---------------------------------------------------------------------------------------------------------------------
// Trajectory: the semantics of utils/src/TrajectoryManager.cpp for trajectories handed over as arrays (MAT-file reading
// stays with the caller): linear up-sampling by ceil(des_fps / fps) samples per interval, dropping the last original
// sample, only when fps != des_fps and there is more than one sample (:23-39,121-126); one cursor for all tracks, clamped
// at the longest track's last sample (:142-153).  Same two calls as the reference's class, so a maintainer can pass the
// reference's own TrajectoryManager instead.
// ---------------------------------------------------------------------------------------------------------------------
class Trajectory {
public:
    void add(const std::string& name, const double* samples, int n, int dim, int fps, int desFps) {
        Track t;
        t.dim = dim;
        if (fps != desFps && n > 1) {
            const double ratio = static_cast<double>(desFps) / fps;
            for (int i = 0; i + 1 < n; ++i)
                for (int k = 0; k < ratio; ++k)
                    for (int d = 0; d < dim; ++d)
                        t.v.push_back(samples[i * dim + d] + (samples[(i + 1) * dim + d] - samples[i * dim + d]) * (k / ratio));
        } else {
            t.v.assign(samples, samples + size_t(n) * dim);
        }
        t.n = int(t.v.size()) / dim;
        if (t.n > m_size) m_size = t.n;
        m_tracks[name] = t;
    }
    bool advanceTrajectory() {
        if (m_index < m_size - 1) ++m_index;
        return true;
    }
    // pointer to the current sample (dim doubles); a track shorter than the cursor holds its last sample
    const double* getCurrentValue(const std::string& name) const {
        const Track& t = m_tracks.at(name);
        const int i = m_index < t.n ? m_index : t.n - 1;
        return t.v.data() + size_t(i) * t.dim;
    }
    bool has(const std::string& name) const { return m_tracks.count(name) != 0; }
    int getTrajectoryIndex() const { return m_index; }

private:
    struct Track { std::vector<double> v; int n = 0, dim = 1; };
    std::map<std::string, Track> m_tracks;
    int m_size = 0, m_index = 0;
};

// ---------------------------------------------------------------------------------------------------------------------
// Snapshots of what the path reads in one tick.
// ---------------------------------------------------------------------------------------------------------------------
struct RobotView {
    int nJoints = 0, nJets = 0;
    double totalMass = 0.0;
    double comPos[3] = {}, basePos[3] = {}, baseRot[9] = {}, baseAngVel[3] = {}, momentumBody[6] = {}, gravity[3] = {};
    double massMatrixBase[36] = {};        // getMassMatrix().block(0, 0, 6, 6), row-major
    double amomBody[24] = {};              // getMatrixAmomJets(true), 6 x 4 row-major
    std::vector<double> jetThrusts;        // nJets
    std::vector<double> jointPos;          // nJoints
    std::vector<double> jetAxes, jetArms;  // nJets x 3
    std::vector<double> relJac;            // getRelativeJacobianJetsBodyFrame()[i], nJets x 6 x nJoints
    std::vector<double> jacJetsLin;        // getJacobian(jet i).topRightCorner(3, nJoints), nJets x 3 x nJoints
    std::vector<double> jacCoM;            // getJacobianCoM().topRightCorner(3, nJoints), 3 x nJoints
    std::vector<std::string> jointNames;   // getJointName(i); read at configure time only
};

struct QPView {
    double throttleMPC[4] = {}, thrustDesMPC[4] = {}, thrustDotDesMPC[4] = {}, estimatedThrustDot[4] = {};
    double posCoMReference[3] = {}, rpyReference[3] = {};
    std::vector<double> outputQPJointsPosition;   // nJoints
};

// what update() writes back into the QPInput (costsVSMPC.cpp:155-160, systemDynamicsVSMPC.cpp:310)
struct QPWriteBack {
    bool referencesPushed = false;
    double posCoMReference[3] = {}, rpyReference[3] = {}, momentumReference[6] = {};
    double alphaGravity = 0.0;
};

// readRobot: through the getters of utils/include/Robot.h (the reference's class or anything with the same members).
// `jacobians` = the quantities only getRobotReference() is asked for every tick (systemDynamicsVSMPC.cpp:159-226,321-350).
template <class RobotT>
void readRobot(RobotT& r, RobotView& v, bool jacobians, bool names = false) {
    v.nJoints = int(r.getNJoints());
    v.nJets = int(r.getNJets());
    v.totalMass = r.getTotalMass();
    const auto pose = r.getBasePose();
    const auto vel = r.getBaseVel();
    const auto com = r.getPositionCoM();
    const auto mom = r.getMomentum(true);
    const auto& grav = r.getGravity();
    for (int i = 0; i < 3; ++i) {
        v.comPos[i] = com(i);
        v.basePos[i] = pose.getPosition()(i);
        v.baseAngVel[i] = vel.getAngularVec3()(i);
        v.gravity[i] = grav(i);
        for (int j = 0; j < 3; ++j) v.baseRot[3 * i + j] = pose.getRotation()(i, j);
    }
    for (int i = 0; i < 6; ++i) v.momentumBody[i] = mom(i);
    const auto thr = r.getJetThrusts();
    v.jetThrusts.resize(v.nJets);
    for (int i = 0; i < v.nJets; ++i) v.jetThrusts[i] = thr(i);
    const auto q = r.getJointPos();
    v.jointPos.resize(v.nJoints);
    for (int i = 0; i < v.nJoints; ++i) v.jointPos[i] = q(i);
    const auto M = r.getMassMatrix();
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) v.massMatrixBase[6 * i + j] = M(i, j);
    const auto Am = r.getMatrixAmomJets(true);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 4; ++j) v.amomBody[4 * i + j] = Am(i, j);
    const auto axes = r.getMatrixOfJetAxes();
    const auto& arms = r.getMatrixOfJetArms();
    v.jetAxes.resize(size_t(3) * v.nJets);
    v.jetArms.resize(size_t(3) * v.nJets);
    for (int i = 0; i < v.nJets; ++i)
        for (int k = 0; k < 3; ++k) { v.jetAxes[3 * i + k] = axes[i](k); v.jetArms[3 * i + k] = arms[i](k); }
    const auto& rel = r.getRelativeJacobianJetsBodyFrame();
    v.relJac.resize(size_t(6) * v.nJets * v.nJoints);
    for (int i = 0; i < v.nJets; ++i)
        for (int a = 0; a < 6; ++a)
            for (int c = 0; c < v.nJoints; ++c) v.relJac[(size_t(i) * 6 + a) * v.nJoints + c] = rel[i](a, c);
    if (jacobians) {
        const auto& jets = r.getJetsList();
        v.jacJetsLin.resize(size_t(3) * v.nJets * v.nJoints);
        for (int i = 0; i < v.nJets; ++i) {
            const auto J = r.getJacobian(jets[i]);
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < v.nJoints; ++c) v.jacJetsLin[(size_t(i) * 3 + a) * v.nJoints + c] = J(a, 6 + c);
        }
        const auto Jc = r.getJacobianCoM();
        v.jacCoM.resize(size_t(3) * v.nJoints);
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < v.nJoints; ++c) v.jacCoM[size_t(a) * v.nJoints + c] = Jc(a, 6 + c);
    }
    if (names) {
        v.jointNames.resize(v.nJoints);
        for (int i = 0; i < v.nJoints; ++i) v.jointNames[i] = r.getJointName(i);
    }
}

template <class QPInputT>
void readQPInput(QPInputT& qp, int nJoints, QPView& v) {   // utils/include/QPInput.h:12-124
    const auto thr = qp.getThrottleMPC();
    const auto td = qp.getThrustDesMPC();
    const auto tdd = qp.getThrustDotDesMPC();
    const auto est = qp.getEstimatedThrustDot();
    for (int i = 0; i < 4; ++i) {
        v.throttleMPC[i] = thr(i);
        v.thrustDesMPC[i] = td(i);
        v.thrustDotDesMPC[i] = tdd(i);
        v.estimatedThrustDot[i] = est(i);
    }
    const auto pr = qp.getPosCoMReference();
    const auto rr = qp.getRPYReference();
    for (int i = 0; i < 3; ++i) { v.posCoMReference[i] = pr(i); v.rpyReference[i] = rr(i); }
    const auto qj = qp.getOutputQPJointsPosition();
    v.outputQPJointsPosition.resize(nJoints);
    for (int i = 0; i < nJoints; ++i) v.outputQPJointsPosition[i] = qj(i);
}

// ---------------------------------------------------------------------------------------------------------------------
// Parameters of group VS_MPC_CONFIG (src/config/vs_mcp_config.xml:5-45) beyond the numeric ones of vsmpc_config.
// ---------------------------------------------------------------------------------------------------------------------
struct MPCParameters {
    vsmpc_config cfg{};
    bool useEstimatedThrust = true;               // vs_mcp_config.xml:20
    bool constantLambda = false;                  // jointsLambdaOption: "unfiltered" (shipped) | "constant"
    std::vector<std::string> controlledJoints;    // vs_mcp_config.xml:9
};

// readParameters: through BLF's IParametersHandler::getParameter(name, value) -> bool (IMPCProblem.cpp:20-60 and the
// plugins' readConfigParameters).  Returns false and names the key like the reference's yError lines do.
template <class HandlerT>
bool readParameters(HandlerT& h, MPCParameters& p, std::string* missing = nullptr) {
    auto need = [&](const char* key, auto& value) {
        if (h.getParameter(key, value)) return true;
        if (missing != nullptr && missing->empty()) *missing = key;
        return false;
    };
    auto vec = [&](const char* key, double* out, size_t n) {
        std::vector<double> v;
        if (!need(key, v)) return false;
        if (v.size() != n) {
            if (missing != nullptr && missing->empty()) *missing = std::string(key) + " (size)";
            return false;
        }
        for (size_t i = 0; i < n; ++i) out[i] = v[i];
        return true;
    };
    bool ok = true, useJet = true;
    vsmpc_config& c = p.cfg;
    ok = need("nIter", c.n_iter) && ok;
    ok = need("nIterSmall", c.n_iter_small) && ok;
    ok = need("controlHorizon", c.control_horizon) && ok;
    ok = need("useJetDynamic", useJet) && ok;
    c.use_jet_dynamic = useJet ? 1 : 0;
    ok = need("periodMPC", c.period_mpc) && ok;
    ok = need("periodMPCSmallSteps", c.period_small) && ok;
    ok = need("periodMPCLargeSteps", c.period_large) && ok;
    ok = vec("weightCoMPos", c.w_com_pos, 3) && ok;
    ok = vec("weightCoMPosError", c.w_com_pos_err, 3) && ok;
    ok = vec("weightLinMom", c.w_lin_mom, 3) && ok;
    ok = vec("weightRPY", c.w_rpy, 3) && ok;
    ok = vec("weightRPYError", c.w_rpy_err, 3) && ok;
    ok = vec("weightAngMom", c.w_ang_mom, 3) && ok;
    ok = vec("weightDeltaJoint", c.w_delta_joint, 8) && ok;
    ok = need("weightThrottle", c.w_throttle) && ok;
    ok = need("weightInitialThrottle", c.w_initial_throttle) && ok;
    ok = need("weightRegularizationJointPos", c.w_reg_joint_pos) && ok;
    ok = need("throttleMin", c.throttle_min) && ok;
    ok = need("throttleMax", c.throttle_max) && ok;
    ok = need("useEstimatedThrust", p.useEstimatedThrust) && ok;
    ok = need("controlledJoints", p.controlledJoints) && ok;
    std::string opt;
    if (need("jointsLambdaOption", opt)) {
        if (opt == "constant") p.constantLambda = true;
        else if (opt == "unfiltered") p.constantLambda = false;
        else { ok = false; if (missing != nullptr && missing->empty()) *missing = "jointsLambdaOption (unfiltered | constant)"; }
    } else {
        ok = false;
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------
// TickMachine: per-instance state + packer + solution bookkeeping.  TrajT: advanceTrajectory() / getCurrentValue(name)
// returning something indexable with [] (Trajectory above, or an adapter over the reference's TrajectoryManager).
// ---------------------------------------------------------------------------------------------------------------------
template <class TrajT = Trajectory>
class TickMachineT {
public:
    // IMPCProblem::configure (IMPCProblem.cpp:3-148): sizes, plugin members, ONE evaluation of every cost and constraint.
    // `position` carries positionCoM / velocityCoM / RPY / RPYDot at 1 / periodMPCLargeSteps (costsVSMPC.cpp:68), `alpha`
    // carries alphaGravity at int(1 / periodMPC) (systemDynamicsVSMPC.cpp:272).
    int configure(const MPCParameters& p, vsmpc_handle* h, const RobotView& robot, const RobotView& ref, QPView& qp,
                  std::shared_ptr<TrajT> position, std::shared_ptr<TrajT> alpha, QPWriteBack& wb) {
        if (robot.nJoints != kRobotJoints || robot.nJets != kJets || ref.nJoints != kRobotJoints || ref.nJets != kJets)
            return VSMPC_ERR_UNSUPPORTED_CONFIG;
        if (int(p.controlledJoints.size()) != VSMPC_N_JOINTS || int(robot.jointNames.size()) != robot.nJoints)
            return VSMPC_ERR_INVALID_ARG;
        m_p = p;
        m_h = h;
        m_pos = position;
        m_alpha = alpha;
        m_nIn = vsmpc_input_doubles(h);
        m_nVar = vsmpc_num_variables(h);
        m_nRef = p.cfg.n_iter - p.cfg.n_iter_small + 1;
        m_ratio = int(std::lround(p.cfg.period_large / p.cfg.period_small));          // constraintsVSMPC.cpp:322
        // joint selection by name (variableSamplingMPC.cpp:47-58, costsVSMPC.cpp:539-550, systemDynamicsVSMPC.cpp:57-66)
        m_sel.clear();
        for (const std::string& name : p.controlledJoints)
            for (int i = 0; i < robot.nJoints; ++i)
                if (name == robot.jointNames[i]) m_sel.push_back(i);
        if (int(m_sel.size()) != VSMPC_N_JOINTS) return VSMPC_ERR_INVALID_ARG;
        int rc = vsmpc_set_kinematics_options(h, m_sel.data(), p.constantLambda ? 1 : 0);
        if (rc != VSMPC_OK) return rc;
        // configure-time members of the plugins
        for (int i = 0; i < 3; ++i) m_initialCoMPos[i] = robot.comPos[i];              // costsVSMPC.cpp:101
        rpyOfRotation(robot.baseRot, m_initialRPY);                                    // costsVSMPC.cpp:102
        rpyOfRotation(ref.baseRot, m_rpyInit);                                         // systemDynamicsVSMPC.cpp:67
        for (int i = 0; i < 3; ++i) { m_rpyOld[i] = m_initialRPY[i]; m_nTurns[i] = 0.0; }   // constraintsVSMPC.cpp:196-199
        m_jointsPositionReference = robot.jointPos;                                    // variableSamplingMPC.cpp:59-60
        m_jointPosReference.resize(VSMPC_N_JOINTS);
        for (int i = 0; i < VSMPC_N_JOINTS; ++i) m_jointPosReference[i] = robot.jointPos[m_sel[i]];   // costsVSMPC.cpp:539-550
        if (p.constantLambda) {                                                        // systemDynamicsVSMPC.cpp:53-55,276-277
            m_relJacInit = robot.relJac;
            m_axesInit = robot.jetAxes;
            m_armsInit = robot.jetArms;
        }
        m_x.assign(m_nVar, 0.0);          // per-tick scratch: allocated here, never in assemble() / solve()
        m_kin.assign(VSMPC_KIN_SIZE, 0.0);
        m_kinPending = false;
        m_window.assign(size_t(12) * m_nRef, 0.0);
        double col[12];
        rc = referenceColumn(robot, col);
        if (rc != VSMPC_OK) return rc;
        for (int c = 0; c < m_nRef; ++c) std::memcpy(&m_window[size_t(12) * c], col, sizeof(col));   // costsVSMPC.cpp:103-113
        m_refCounter = m_ratio - 1;                                                    // costsVSMPC.cpp:118
        m_throttleCounter = m_ratio - 1;                                               // constraintsVSMPC.cpp:335
        m_record.assign(m_nIn, 0.0);
        m_QPSolution.assign(m_nVar, 0.0);
        m_deltaJoints.assign(VSMPC_N_JOINTS, 0.0);
        m_thrust.assign(4, 0.0);
        m_thrustDot.assign(4, 0.0);
        m_throttle.assign(4, 0.0);
        m_finalState.assign(VSMPC_N_STATES, 0.0);
        m_status = 0;
        return assemble(robot, ref, qp, wb);   // IMPCProblem.cpp:80-132 evaluates every plugin once
    }

    // Fused tick (default): assemble() only PACKS -- the raw Robot quantities of the Lambda / I_G terms go into the
    // kinematics record -- and solve() submits kinematics -> solve in one vsmpc_tick call (one synchronisation per tick).
    // Off: assemble() completes the record itself (vsmpc_kinematics_batch) and solve() is vsmpc_solve_batch: two round
    // trips, but record() is complete between update() and solveMPC().
    void setFusedTick(bool fused) { m_fused = fused; }
    bool fusedTick() const { return m_fused; }

    // One IMPCProblem::update worth of plugin evaluations, in the reference's order -- costs (reference tracking,
    // regularisation, throttle anchor, joint posture), then constraints (dynamics: angular, linear, jets; initial state;
    // throttle box) (IMPCProblem.cpp:150-194, variableSamplingMPC.cpp:70-84) -- into the vsmpc_input record.
    int assemble(const RobotView& robot, const RobotView& ref, QPView& qp, QPWriteBack& wb) {
        double* rec = m_record.data();
        std::fill(m_record.begin(), m_record.end(), 0.0);
        wb.referencesPushed = false;
        // --- ReferenceTrackingCost::computeHessianAndGradient (costsVSMPC.cpp:121-165)
        if (m_refCounter == m_ratio - 1) {
            m_pos->advanceTrajectory();
            double col[12];
            const int rc = referenceColumn(robot, col);
            if (rc != VSMPC_OK) return rc;
            std::memmove(&m_window[0], &m_window[12], sizeof(double) * 12 * size_t(m_nRef - 1));
            std::memcpy(&m_window[size_t(12) * (m_nRef - 1)], col, sizeof(col));
            for (int i = 0; i < 3; ++i) {
                qp.posCoMReference[i] = wb.posCoMReference[i] = m_window[i];
                qp.rpyReference[i] = wb.rpyReference[i] = m_window[6 + i];
                wb.momentumReference[i] = m_window[3 + i];
                wb.momentumReference[3 + i] = m_window[9 + i];
            }
            wb.referencesPushed = true;
            m_refCounter = 0;
        } else {
            ++m_refCounter;
        }
        std::memcpy(rec + VSMPC_IN_XREF, m_window.data(), sizeof(double) * m_window.size());
        // --- JointPositionRegularizationCost (costsVSMPC.cpp:574-589): selected by name
        for (int i = 0; i < VSMPC_N_JOINTS; ++i)
            rec[VSMPC_IN_QERR + i] = qp.outputQPJointsPosition[m_sel[i]] - m_jointPosReference[i];
        // --- dynamics.  wR_b, the Jacobians, masses and axes come from getRobotReference(); omega and the jets'
        //     linearisation thrust from getRobot() (systemDynamicsVSMPC.cpp:107-108,324-325,401-404)
        const double* R = ref.baseRot;
        rec[VSMPC_IN_MASS] = double(float(ref.totalMass));                             // Robot.h:338 keeps a float
        std::memcpy(rec + VSMPC_IN_WRB, R, sizeof(double) * 9);
        mulRt(R, robot.baseAngVel, rec + VSMPC_IN_OMEGA);
        double alpha;
        {
            const auto a = m_alpha->getCurrentValue(std::string("alphaGravity"));      // systemDynamicsVSMPC.cpp:308-311
            alpha = a[0];
        }
        wb.alphaGravity = alpha;
        m_alpha->advanceTrajectory();
        rec[VSMPC_IN_ALPHA] = alpha;
        std::memcpy(rec + VSMPC_IN_GRAV, ref.gravity, sizeof(double) * 3);
        std::memcpy(rec + VSMPC_IN_AMOM, ref.amomBody, sizeof(double) * 24);
        kinematicsRecord(robot, ref, m_kin.data());
        m_kinPending = m_fused;
        if (!m_fused) {
            double out[VSMPC_KIN_OUT];
            const int rc = vsmpc_kinematics_batch(m_h, m_kin.data(), 1, out, nullptr);
            if (rc != VSMPC_OK) return rc;
            std::memcpy(rec + VSMPC_IN_LLIN, out, sizeof(double) * 24);
            std::memcpy(rec + VSMPC_IN_LANG, out + 24, sizeof(double) * 24);
            std::memcpy(rec + VSMPC_IN_INERTIA, out + 48, sizeof(double) * 9);
        }
        rpyOfRotation(R, rec + VSMPC_IN_RPY);
        std::memcpy(rec + VSMPC_IN_PREF, qp.posCoMReference, sizeof(double) * 3);      // :316 (after the cost's push)
        std::memcpy(rec + VSMPC_IN_RPYINIT, m_rpyInit, sizeof(double) * 3);
        for (int i = 0; i < 4; ++i) {                                                  // :401-409
            rec[VSMPC_IN_T0 + i] = m_p.useEstimatedThrust ? robot.jetThrusts[i] : qp.thrustDesMPC[i];
            rec[VSMPC_IN_TD0 + i] = m_p.useEstimatedThrust ? qp.estimatedThrustDot[i] : qp.thrustDotDesMPC[i];
            rec[VSMPC_IN_UPREV + i] = qp.throttleMPC[i];
            rec[VSMPC_IN_TDES + i] = qp.thrustDesMPC[i];
            rec[VSMPC_IN_TDDES + i] = qp.thrustDotDesMPC[i];
        }
        // --- ConstraintInitialState (constraintsVSMPC.cpp:206-247): getRobot()
        double rpy[3], un[3];
        rpyOfRotation(robot.baseRot, rpy);
        for (int i = 0; i < 3; ++i) {
            if (rpy[i] - m_rpyOld[i] > kPi) m_nTurns[i] -= 1.0;
            else if (rpy[i] - m_rpyOld[i] < -kPi) m_nTurns[i] += 1.0;
            un[i] = rpy[i] + 2.0 * kPi * m_nTurns[i];
            m_rpyOld[i] = rpy[i];
        }
        double* x0 = rec + VSMPC_IN_X0;
        for (int i = 0; i < 3; ++i) {
            x0[i] = robot.comPos[i];
            x0[3 + i] = robot.momentumBody[i];
            x0[6 + i] = un[i];
            x0[9 + i] = robot.momentumBody[3 + i];
            x0[20 + i] = robot.comPos[i] - qp.posCoMReference[i];
            x0[23 + i] = un[i] - qp.rpyReference[i];
        }
        for (int i = 0; i < 4; ++i) {
            x0[12 + i] = m_p.useEstimatedThrust ? robot.jetThrusts[i] : qp.thrustDesMPC[i];
            x0[16 + i] = m_p.useEstimatedThrust ? qp.estimatedThrustDot[i] : qp.thrustDotDesMPC[i];
        }
        // --- ThrottleConstraint (constraintsVSMPC.cpp:351-372)
        rec[VSMPC_IN_HOLD] = (m_throttleCounter != m_ratio - 1) ? 1.0 : 0.0;
        m_throttleCounter = (m_throttleCounter == m_ratio - 1) ? 0 : m_throttleCounter + 1;
        return VSMPC_OK;
    }

    // VariableSamplingMPC::solveMPC (variableSamplingMPC.cpp:88-112) on the record of the last assemble() / setRecord()
    int solve() {
        double fm[VSMPC_FM_SIZE];
        std::vector<double>& x = m_x;
        int status = 0;
        const int rc = m_kinPending ? vsmpc_tick(m_h, m_kin.data(), m_record.data(), 1, x.data(), fm, &status, nullptr, nullptr)
                                    : vsmpc_solve_batch(m_h, m_record.data(), 1, x.data(), fm, &status, nullptr, nullptr);
        m_kinPending = false;
        m_status = (rc == VSMPC_OK) ? status : VSMPC_STATUS_NUMERICAL;
        if (m_status == VSMPC_STATUS_SOLVED) {                                         // :91: consume only if Solved
            m_QPSolution.swap(x);                                                      // (both stay m_nVar long)
            for (int i = 0; i < VSMPC_N_JOINTS; ++i) m_deltaJoints[i] = fm[VSMPC_FM_DQ + i];
            for (int i = 0; i < 4; ++i) {
                m_throttle[i] = fm[VSMPC_FM_THROTTLE + i];
                m_thrust[i] = fm[VSMPC_FM_THRUST + i];
                m_thrustDot[i] = fm[VSMPC_FM_THRUSTDOT + i];
            }
            const int N = m_p.cfg.n_iter;
            for (int i = 0; i < VSMPC_N_STATES; ++i) m_finalState[i] = m_QPSolution[size_t(VSMPC_N_STATES) * N + i];
            for (int i = 0; i < VSMPC_N_JOINTS; ++i) m_jointsPositionReference[m_sel[i]] += m_deltaJoints[i];   // :104-108
        }
        return rc;
    }

    // record-level front end: the caller delivers the record (and, optionally, lets the hold / unwrap state machine run)
    void initRecordLevel(const vsmpc_config& cfg, vsmpc_handle* h, const double* jointPos, const double rpy0[3]) {
        m_p = MPCParameters{};
        m_p.cfg = cfg;
        m_h = h;
        m_nIn = vsmpc_input_doubles(h);
        m_nVar = vsmpc_num_variables(h);
        m_ratio = int(std::lround(cfg.period_large / cfg.period_small));
        m_sel.resize(VSMPC_N_JOINTS);
        for (int i = 0; i < VSMPC_N_JOINTS; ++i) m_sel[i] = kJointOffset + i;
        m_jointsPositionReference.assign(kRobotJoints, 0.0);
        if (jointPos != nullptr) m_jointsPositionReference.assign(jointPos, jointPos + kRobotJoints);
        m_throttleCounter = 0;   // starts at ratio - 1 (constraintsVSMPC.cpp:335) and is consumed once by configure
        for (int i = 0; i < 3; ++i) { m_rpyOld[i] = rpy0 != nullptr ? rpy0[i] : 0.0; m_nTurns[i] = 0.0; }
        m_record.assign(m_nIn, 0.0);
        m_QPSolution.assign(m_nVar, 0.0);
        m_deltaJoints.assign(VSMPC_N_JOINTS, 0.0);
        m_thrust.assign(4, 0.0);
        m_thrustDot.assign(4, 0.0);
        m_throttle.assign(4, 0.0);
        m_finalState.assign(VSMPC_N_STATES, 0.0);
        m_x.assign(m_nVar, 0.0);
        m_kinPending = false;
        m_status = 0;
    }
    // the two counters of the record-level front end alone (no handle needed)
    void initCounters(const vsmpc_config& cfg, const double rpy0[3]) {
        m_ratio = int(std::lround(cfg.period_large / cfg.period_small));
        m_throttleCounter = 0;
        for (int i = 0; i < 3; ++i) { m_rpyOld[i] = rpy0 != nullptr ? rpy0[i] : 0.0; m_nTurns[i] = 0.0; }
    }
    void setRecord(const double* record, bool applyTickState) {
        m_record.assign(record, record + m_nIn);   // (same size as before: no reallocation)
        m_kinPending = false;
        if (!applyTickState) return;
        m_record[VSMPC_IN_HOLD] = nextHoldFlag() ? 1.0 : 0.0;
        double un[3];
        unwrapRPY(&m_record[VSMPC_IN_RPY], un);
        for (int i = 0; i < 3; ++i) {
            const double refv = m_record[VSMPC_IN_X0 + 6 + i] - m_record[VSMPC_IN_X0 + 23 + i];   // RPYReference
            m_record[VSMPC_IN_X0 + 6 + i] = un[i];                                                 // constraintsVSMPC.cpp:212
            m_record[VSMPC_IN_X0 + 23 + i] = un[i] - refv;                                         // :227-228
        }
    }
    bool nextHoldFlag() {                                                              // constraintsVSMPC.cpp:351,366-372
        const bool hold = m_throttleCounter != m_ratio - 1;
        m_throttleCounter = (m_throttleCounter == m_ratio - 1) ? 0 : m_throttleCounter + 1;
        return hold;
    }
    void unwrapRPY(const double rpy[3], double out[3]) {                               // constraintsVSMPC.cpp:232-247
        for (int i = 0; i < 3; ++i) {
            if (rpy[i] - m_rpyOld[i] > kPi) m_nTurns[i] -= 1.0;
            else if (rpy[i] - m_rpyOld[i] < -kPi) m_nTurns[i] += 1.0;
            out[i] = rpy[i] + 2.0 * kPi * m_nTurns[i];
            m_rpyOld[i] = rpy[i];
        }
    }

    // state
    const std::vector<double>& record() const { return m_record; }
    const std::vector<double>& solution() const { return m_QPSolution; }
    const std::vector<double>& jointsPositionReference() const { return m_jointsPositionReference; }
    const std::vector<double>& throttle() const { return m_throttle; }
    const std::vector<double>& thrust() const { return m_thrust; }
    const std::vector<double>& thrustDot() const { return m_thrustDot; }
    const std::vector<double>& finalState() const { return m_finalState; }
    const std::vector<int>& jointSelector() const { return m_sel; }
    const double* nTurns() const { return m_nTurns; }
    int status() const { return m_status; }
    int nVar() const { return m_nVar; }
    int nIn() const { return m_nIn; }
    const MPCParameters& parameters() const { return m_p; }

private:
    // one window column from the CURRENT trajectory sample, getRobot()'s attitude and mass (costsVSMPC.cpp:105-112,127-146)
    int referenceColumn(const RobotView& robot, double col[12]) {
        // getCurrentValue may return a pointer (Trajectory) or an Eigen::VectorXd BY VALUE (the reference's
        // TrajectoryManager, TrajectoryManager.h:104): keep what it returns alive, read with []
        const auto p = m_pos->getCurrentValue(std::string("positionCoM"));
        const auto v = m_pos->getCurrentValue(std::string("velocityCoM"));
        double mv[3];
        for (int i = 0; i < 3; ++i) { col[i] = m_initialCoMPos[i] + p[i]; mv[i] = robot.totalMass * v[i]; }
        mulRt(robot.baseRot, mv, col + 3);
        double rpyTrack[3] = {0, 0, 0}, rpyDot[3] = {0, 0, 0};
        if (hasTrack(*m_pos, "RPY")) {
            const auto t = m_pos->getCurrentValue(std::string("RPY"));
            for (int i = 0; i < 3; ++i) rpyTrack[i] = t[i];
        }
        if (hasTrack(*m_pos, "RPYDot")) {
            const auto t = m_pos->getCurrentValue(std::string("RPYDot"));
            for (int i = 0; i < 3; ++i) rpyDot[i] = t[i];
        }
        for (int i = 0; i < 3; ++i) { col[6 + i] = m_initialRPY[i] + rpyTrack[i]; col[9 + i] = 0.0; }
        if (rpyDot[0] != 0.0 || rpyDot[1] != 0.0 || rpyDot[2] != 0.0) {
            // m_inertia * m_W * RPYDot with getRobot()'s locked inertia (costsVSMPC.cpp:111-112,143-146,266-286); I_G on the
            // device, from a kinematics record that carries only the quantities I_G needs
            std::vector<double>& kin = m_kin;          // (rewritten in full by kinematicsRecord() later in this tick)
            std::fill(kin.begin(), kin.end(), 0.0);
            double out[VSMPC_KIN_OUT];
            std::memcpy(&kin[VSMPC_KIN_WRB], robot.baseRot, sizeof(double) * 9);
            std::memcpy(&kin[VSMPC_KIN_MB], robot.massMatrixBase, sizeof(double) * 36);
            for (int i = 0; i < 3; ++i) kin[VSMPC_KIN_R + i] = robot.comPos[i] - robot.basePos[i];
            const int rc = vsmpc_kinematics_batch(m_h, kin.data(), 1, out, nullptr);
            if (rc != VSMPC_OK) return rc;
            double rpy[3];
            rpyOfRotation(robot.baseRot, rpy);
            const double W[9] = {1.0, 0.0, -std::sin(rpy[1]),
                                 0.0, std::cos(rpy[0]), std::cos(rpy[1]) * std::sin(rpy[0]),
                                 0.0, -std::sin(rpy[0]), std::cos(rpy[0]) * std::cos(rpy[1])};
            double wd[3];
            for (int i = 0; i < 3; ++i) wd[i] = W[3 * i] * rpyDot[0] + W[3 * i + 1] * rpyDot[1] + W[3 * i + 2] * rpyDot[2];
            const double* IG = out + 48;
            for (int i = 0; i < 3; ++i) col[9 + i] = IG[3 * i] * wd[0] + IG[3 * i + 1] * wd[1] + IG[3 * i + 2] * wd[2];
        }
        return VSMPC_OK;
    }

    // raw Robot quantities in the VSMPC_KIN_* layout (systemDynamicsVSMPC.cpp:128-130,159-226,321-350)
    void kinematicsRecord(const RobotView& robot, const RobotView& ref, double* k) const {
        const int nJ = kRobotJoints;
        std::memset(k, 0, sizeof(double) * VSMPC_KIN_SIZE);
        std::memcpy(k + VSMPC_KIN_WRB, ref.baseRot, sizeof(double) * 9);
        for (int i = 0; i < 4; ++i) k[VSMPC_KIN_THRUST + i] = ref.jetThrusts[i];
        const bool cst = m_p.constantLambda;
        const std::vector<double>& axes = cst ? m_axesInit : ref.jetAxes;
        const std::vector<double>& arms = cst ? m_armsInit : ref.jetArms;
        const std::vector<double>& rel = cst ? m_relJacInit : ref.relJac;
        std::memcpy(k + VSMPC_KIN_AXES, axes.data(), sizeof(double) * 12);
        std::memcpy(k + VSMPC_KIN_ARMS, arms.data(), sizeof(double) * 12);
        for (int i = 0; i < 4; ++i)
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < nJ; ++c) {
                    k[VSMPC_KIN_JREL + (i * 3 + a) * nJ + c] = rel[(size_t(i) * 6 + 3 + a) * nJ + c];   // bottomRows(3)
                    k[VSMPC_KIN_JFRAME + (i * 3 + a) * nJ + c] =
                        cst ? rel[(size_t(i) * 6 + a) * nJ + c] : ref.jacJetsLin[(size_t(i) * 3 + a) * nJ + c];
                }
        if (cst) {
            for (int i = 0; i < 4; ++i) k[VSMPC_KIN_JCOM + i] = robot.jetThrusts[i];   // systemDynamicsVSMPC.cpp:196
        } else {
            std::memcpy(k + VSMPC_KIN_JCOM, ref.jacCoM.data(), sizeof(double) * 3 * nJ);
        }
        std::memcpy(k + VSMPC_KIN_MB, ref.massMatrixBase, sizeof(double) * 36);
        for (int i = 0; i < 3; ++i) k[VSMPC_KIN_R + i] = ref.comPos[i] - ref.basePos[i];
    }

    MPCParameters m_p;
    vsmpc_handle* m_h = nullptr;
    std::shared_ptr<TrajT> m_pos, m_alpha;
    int m_nIn = 0, m_nVar = 0, m_nRef = 0, m_ratio = 20, m_refCounter = 0, m_throttleCounter = 0, m_status = 0;
    std::vector<int> m_sel;
    double m_initialCoMPos[3] = {}, m_initialRPY[3] = {}, m_rpyInit[3] = {}, m_rpyOld[3] = {}, m_nTurns[3] = {};
    std::vector<double> m_window, m_record, m_QPSolution, m_jointsPositionReference, m_jointPosReference, m_deltaJoints;
    std::vector<double> m_thrust, m_thrustDot, m_throttle, m_finalState;
    std::vector<double> m_relJacInit, m_axesInit, m_armsInit;
    std::vector<double> m_x, m_kin;     // per-tick scratch (solution before it is consumed; kinematics record)
    bool m_fused = true, m_kinPending = false;
};
using TickMachine = TickMachineT<Trajectory>;

// ---------------------------------------------------------------------------------------------------------------------
// getters shared by the front ends: size-checked like the reference (variableSamplingMPC.cpp:114-217), bool return,
// writing through operator() / size() (Eigen::Ref<Eigen::VectorXd> in the reference).
// ---------------------------------------------------------------------------------------------------------------------
template <class OutT>
bool copyOut(const std::vector<double>& v, size_t off, size_t n, OutT&& out) {
    if (size_t(out.size()) != n) return false;
    for (size_t i = 0; i < n; ++i) out(i) = v[off + i];
    return true;
}
struct SpanOut {   // plain pointer + size with the same two members
    double* p; int n;
    int size() const { return n; }
    double& operator()(size_t i) { return p[i]; }
};

// ---------------------------------------------------------------------------------------------------------------------
// The reference's surface.  ParamsT: BLF IParametersHandler-like (getParameter(name, value) -> bool); QPInputT / RobotT:
// utils/include/QPInput.h / Robot.h-like; TrajT as above.  `trajectories(handler, position, alpha)` is how the caller
// supplies the two trajectory sets (the reference reads them from MAT files named in groups POSITION_TRAJECTORY and
// TRAJECTORY_MANAGER, costsVSMPC.cpp:36-68, systemDynamicsVSMPC.cpp:263-272).
// ---------------------------------------------------------------------------------------------------------------------
template <class QPInputT, class TrajT = Trajectory>
class VariableSamplingMPCT {
public:
    VariableSamplingMPCT() = default;
    ~VariableSamplingMPCT() { if (m_h) vsmpc_destroy(m_h); }
    VariableSamplingMPCT(const VariableSamplingMPCT&) = delete;
    VariableSamplingMPCT& operator=(const VariableSamplingMPCT&) = delete;

    void setTrajectories(std::shared_ptr<TrajT> position, std::shared_ptr<TrajT> alpha) { m_pos = position; m_alpha = alpha; }
    void setDevice(int device) { m_device = device; }
    // one submission per tick (default) or the two-call form whose record is complete right after update(); before configure
    void setFusedTick(bool fused) { m_fused = fused; }

    // IMPCProblem::configure takes std::weak_ptr<IParametersHandler> (IMPCProblem.h:35; MPCPyBindings.cpp:24-32 makes
    // one from the shared_ptr Python holds): both spellings, then the handler itself
    template <class HandlerT>
    bool configure(std::weak_ptr<HandlerT> parametersHandler, QPInputT& qpInput) {
        auto ptr = parametersHandler.lock();
        if (!ptr) { m_message = "parameters handler expired"; return false; }
        return configure(*ptr, qpInput);
    }
    template <class HandlerT>
    bool configure(std::shared_ptr<HandlerT> parametersHandler, QPInputT& qpInput) {
        if (!parametersHandler) { m_message = "parameters handler is null"; return false; }
        return configure(*parametersHandler, qpInput);
    }
    template <class ParamsT>
    bool configure(ParamsT& parametersHandler, QPInputT& qpInput) {
        MPCParameters p;
        m_message.clear();
        if (!readParameters(parametersHandler, p, &m_message)) {
            m_message = "Parameter '" + m_message + "' not found in the config file.";
            return false;
        }
        if (!m_pos || !m_alpha) { m_message = "trajectories not set"; return false; }
        if (m_h) { vsmpc_destroy(m_h); m_h = nullptr; }
        m_lastError = vsmpc_create(&p.cfg, m_device, 1, &m_h);
        if (m_lastError != VSMPC_OK) { m_message = vsmpc_strerror(m_lastError); return false; }
        RobotView robot, ref;
        QPView qv;
        QPWriteBack wb;
        readRobot(*qpInput.getRobot(), robot, false, true);
        readRobot(*qpInput.getRobotReference(), ref, true);
        readQPInput(qpInput, robot.nJoints, qv);
        m_tick.setFusedTick(m_fused);
        m_lastError = m_tick.configure(p, m_h, robot, ref, qv, m_pos, m_alpha, wb);
        if (m_lastError != VSMPC_OK) { m_message = vsmpc_strerror(m_lastError); return false; }
        writeBack(qpInput, wb);
        m_configured = true;
        return true;
    }

    bool update(QPInputT& qpInput) {   // IMPCProblem::update (IMPCProblem.cpp:150-194)
        if (!m_configured) return false;
        RobotView& robot = m_robot;      // members: their vectors keep their capacity from tick to tick
        RobotView& ref = m_ref;
        QPView& qv = m_qv;
        QPWriteBack wb;
        readRobot(*qpInput.getRobot(), robot, false);
        readRobot(*qpInput.getRobotReference(), ref, true);
        readQPInput(qpInput, robot.nJoints, qv);
        m_lastError = m_tick.assemble(robot, ref, qv, wb);
        writeBack(qpInput, wb);
        return m_lastError == VSMPC_OK;
    }

    bool solveMPC() {                  // variableSamplingMPC.cpp:88-112: true whatever the status
        if (m_configured) m_lastError = m_tick.solve();
        return true;
    }

    int getQPProblemStatus() const { return m_tick.status(); }
    int getLastError() const { return m_lastError; }
    const std::string& getLastMessage() const { return m_message; }
    unsigned int getNOptimizationVariables() const { return static_cast<unsigned int>(m_tick.nVar()); }
    template <class OutT> bool getMPCSolution(OutT&& out) const {
        const size_t off = size_t(VSMPC_N_STATES) * (m_tick.parameters().cfg.n_iter + 1);
        return copyOut(m_tick.solution(), off, m_tick.solution().size() - off, out);
    }
    template <class OutT> bool getJointsReferencePosition(OutT&& out) const {
        return copyOut(m_tick.jointsPositionReference(), 0, m_tick.jointsPositionReference().size(), out);
    }
    template <class OutT> bool getThrottleReference(OutT&& out) const { return copyOut(m_tick.throttle(), 0, 4, out); }
    template <class OutT> bool getThrustReference(OutT&& out) const { return copyOut(m_tick.thrust(), 0, 4, out); }
    template <class OutT> bool getThrustDotReference(OutT&& out) const { return copyOut(m_tick.thrustDot(), 0, 4, out); }
    template <class OutT> bool getFinalCoMPosition(OutT&& out) const { return copyOut(m_tick.finalState(), 0, 3, out); }
    template <class OutT> bool getFinalLinMom(OutT&& out) const { return copyOut(m_tick.finalState(), 3, 3, out); }
    template <class OutT> bool getFinalRPY(OutT&& out) const { return copyOut(m_tick.finalState(), 6, 3, out); }
    template <class OutT> bool getFinalAngMom(OutT&& out) const { return copyOut(m_tick.finalState(), 9, 3, out); }
    double getNStatesMPC() const { return VSMPC_N_STATES; }                       // the reference returns double (:219-227)
    double getNInputMPC() const { return VSMPC_N_JOINTS + VSMPC_N_THRUSTS; }
    const std::vector<double>& getSolution() const { return m_tick.solution(); }  // IMPCProblem::getSolution
    const TickMachineT<TrajT>& tickMachine() const { return m_tick; }

private:
    static void writeBack(QPInputT& qp, const QPWriteBack& wb) {
        if (wb.referencesPushed) {                                                    // costsVSMPC.cpp:155-160
            qp.setPosCoMReference(fixedArg<3>(wb.posCoMReference));                   // QPInput.h:41
            qp.setRPYReference(fixedArg<3>(wb.rpyReference));                         // QPInput.h:45
            qp.setMomentumReference(fixedArg<6>(wb.momentumReference));               // QPInput.h:67
        }
        qp.setAlphaGravity(wb.alphaGravity);                                          // systemDynamicsVSMPC.cpp:310
    }
    vsmpc_handle* m_h = nullptr;
    TickMachineT<TrajT> m_tick;
    std::shared_ptr<TrajT> m_pos, m_alpha;
    RobotView m_robot, m_ref;
    QPView m_qv;
    int m_device = 0, m_lastError = 0;
    bool m_configured = false, m_fused = true;
    std::string m_message;
};

// ---------------------------------------------------------------------------------------------------------------------
// Record-level front end: update(record) with the already extracted VSMPC_IN_* image of the tick.
// ---------------------------------------------------------------------------------------------------------------------
class VariableSamplingMPC {
public:
    VariableSamplingMPC() = default;
    ~VariableSamplingMPC() { if (m_h) vsmpc_destroy(m_h); }
    VariableSamplingMPC(const VariableSamplingMPC&) = delete;
    VariableSamplingMPC& operator=(const VariableSamplingMPC&) = delete;

    bool configure(const vsmpc_config& cfg, const double* initialJointPositions /* 23 or nullptr */,
                   const double initialRPY[3], int device = 0) {
        if (m_h) { vsmpc_destroy(m_h); m_h = nullptr; }
        m_lastError = vsmpc_create(&cfg, device, 1, &m_h);
        if (m_lastError != VSMPC_OK) return false;
        m_cfg = cfg;
        m_tick.initRecordLevel(cfg, m_h, initialJointPositions, initialRPY);
        m_haveRecord = false;
        return true;
    }
    // When `applyTickState` is true the wrapper overwrites the hold flag and the unwrapped RPY entries of X0 from its own
    // state machine (as the reference's plugins would); otherwise the record is used verbatim.
    bool update(const double* record, bool applyTickState = false) {
        if (!m_h || !record) return false;
        m_tick.setRecord(record, applyTickState);
        m_haveRecord = true;
        return true;
    }
    bool solveMPC() {
        if (m_h && m_haveRecord) m_lastError = m_tick.solve();
        return true;
    }
    int getQPProblemStatus() const { return m_tick.status(); }
    int getLastError() const { return m_lastError; }
    unsigned int getNOptimizationVariables() const { return static_cast<unsigned int>(m_tick.nVar()); }
    bool getMPCSolution(double* out, int size) const {
        const size_t off = size_t(VSMPC_N_STATES) * (m_cfg.n_iter + 1);
        return copyOut(m_tick.solution(), off, m_tick.solution().size() - off, SpanOut{out, size});
    }
    bool getJointsReferencePosition(double* out, int size) const {
        return copyOut(m_tick.jointsPositionReference(), 0, m_tick.jointsPositionReference().size(), SpanOut{out, size});
    }
    bool getThrottleReference(double* out, int size) const { return copyOut(m_tick.throttle(), 0, 4, SpanOut{out, size}); }
    bool getThrustReference(double* out, int size) const { return copyOut(m_tick.thrust(), 0, 4, SpanOut{out, size}); }
    bool getThrustDotReference(double* out, int size) const { return copyOut(m_tick.thrustDot(), 0, 4, SpanOut{out, size}); }
    bool getFinalCoMPosition(double* out, int size) const { return copyOut(m_tick.finalState(), 0, 3, SpanOut{out, size}); }
    bool getFinalLinMom(double* out, int size) const { return copyOut(m_tick.finalState(), 3, 3, SpanOut{out, size}); }
    bool getFinalRPY(double* out, int size) const { return copyOut(m_tick.finalState(), 6, 3, SpanOut{out, size}); }
    bool getFinalAngMom(double* out, int size) const { return copyOut(m_tick.finalState(), 9, 3, SpanOut{out, size}); }
    double getNStatesMPC() const { return VSMPC_N_STATES; }
    double getNInputMPC() const { return VSMPC_N_JOINTS + VSMPC_N_THRUSTS; }
    const std::vector<double>& getSolution() const { return m_tick.solution(); }
    TickMachine& tickState() { return m_tick; }
    const vsmpc_config& config() const { return m_cfg; }
    int inputDoubles() const { return m_tick.nIn(); }

private:
    vsmpc_handle* m_h = nullptr;
    vsmpc_config m_cfg{};
    TickMachine m_tick;
    int m_lastError = 0;
    bool m_haveRecord = false;
};

}  // namespace vsmpc_host
#endif  // VARIABLE_SAMPLING_MPC_HPP
