// Drives vsmpc_host::VariableSamplingMPCT<QPInput, TrajectoryManager> -- the reference-side binding of INTEGRATION.md
// section 2, through tests/cpp/integration_snippet.cpp VERBATIM -- against signature-exact stand-ins of the reference's
// headers (tests/cpp/refstub/: Eigen-typed QPInput setters, an Eigen::VectorXd-returning TrajectoryManager without has(),
// a weak_ptr<IParametersHandler>; see its README), fed from a scenario file written by the Python test.  If the header
// assumed anything the reference's declarations do not offer, this file would not BUILD.
//   reference_surface_driver <scenario.bin> <out.bin> [two-call]
//   reference_surface_driver <scenario.bin> - latency <ticks> [two-call]   -> one JSON line: p50 / p99 / mean wall-clock of
//       update(qpInput) + solveMPC() per tick (std::chrono around the two calls only; loading the provider state into the
//       stand-in Robot is the provider's work and outside the timed region); bench.py's latency_reference_surface_us
// Scenario (doubles): header[16] | trajectories | config | controlled-joint indices[8] | initial QPInput | per tick
// (configure-time first): robot block (+ reference-robot block when header says the two differ) + estimatedThrustDot[4].
// Output per tick: record[n_in] | status | thrust[4] | thrustDot[4] | throttle[4] | joints[23] | posCoMRef[3] | rpyRef[3] |
// alpha | momentumRef[6].
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "integration_snippet.cpp"   // the INTEGRATION.md snippet, verbatim (includes VariableSamplingMPC.hpp, QPInput.h, ...)

// compile-time pins of what the stand-ins (and hence the reference) do NOT convert from
static_assert(!std::is_convertible<const double*, Eigen::Ref<const Eigen::Vector3d>>::value, "refstub: double* must not convert to Ref");
static_assert(!std::is_convertible<double (&)[6], const Eigen::Vector6d&>::value, "refstub: double[6] must not convert to Vector6d");
static_assert(std::is_same<decltype(std::declval<const TrajectoryManager&>().getCurrentValue(std::string())), Eigen::VectorXd>::value,
              "getCurrentValue returns by value");

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<double> s(size_t(bytes) / sizeof(double));
    if (std::fread(s.data(), sizeof(double), s.size(), f) != s.size()) return 3;
    std::fclose(f);
    const double* p = s.data();
    const int nTicks = int(p[0]), nPos = int(p[1]), posFps = int(p[2]), nAlpha = int(p[3]), alphaFps = int(p[4]);
    const bool useEst = p[5] != 0.0, constantLambda = p[6] != 0.0, distinct = p[7] != 0.0, dropKey = p[8] != 0.0;
    p += 16;
    const double* posCoM = p; p += 3 * nPos;
    const double* velCoM = p; p += 3 * nPos;
    const double* rpyTrack = p; p += 3 * nPos;
    const double* rpyDotTrack = p; p += 3 * nPos;
    const double* alphaTrack = p; p += nAlpha;
    using BipedalLocomotion::ParametersHandler::YarpImplementation;
    auto handler = std::make_shared<YarpImplementation>();
    YarpImplementation& h = *handler;
    const char* scalarKeys[] = {"nIter", "nIterSmall", "controlHorizon", "useJetDynamic", "periodMPC", "periodMPCSmallSteps",
                                "periodMPCLargeSteps"};
    double periodMPC = 0.0, periodLarge = 0.0;
    for (const char* k : scalarKeys) {
        if (!std::strcmp(k, "periodMPC")) periodMPC = *p;
        if (!std::strcmp(k, "periodMPCLargeSteps")) periodLarge = *p;
        h.testSet(k, *p++);
    }
    (void)periodMPC; (void)periodLarge;
    const char* vecKeys[] = {"weightCoMPos", "weightCoMPosError", "weightLinMom", "weightRPY", "weightRPYError", "weightAngMom"};
    for (const char* k : vecKeys) { h.testSet(k, std::vector<double>(p, p + 3)); p += 3; }
    h.testSet("weightDeltaJoint", std::vector<double>(p, p + 8)); p += 8;
    const char* tailKeys[] = {"weightThrottle", "weightInitialThrottle", "weightRegularizationJointPos", "throttleMin", "throttleMax"};
    for (const char* k : tailKeys) h.testSet(k, *p++);
    h.testSet("useEstimatedThrust", useEst ? 1.0 : 0.0);
    h.testSet("jointsLambdaOption", constantLambda ? "constant" : "unfiltered");
    std::vector<std::string> names;
    for (int i = 0; i < 8; ++i) names.push_back("joint_" + std::to_string(int(*p++)));
    h.testSet("controlledJoints", names);
    if (dropKey) h.testErase("weightThrottle");
    // the two trajectory groups as src/config/vs_mcp_config.xml:34-40 holds them: a file name each.  The stand-in
    // TrajectoryManager takes the arrays registered under that name instead of opening a MAT file.
    {
        auto positionGroup = std::make_shared<YarpImplementation>(), alphaGroup = std::make_shared<YarpImplementation>();
        positionGroup->testSet("trajectoryFile", "src/trajectories/minimumJerkTrajectory.mat");
        alphaGroup->testSet("trajectoryFile", "src/trajectories/alphaGravity.mat");
        h.setGroup("POSITION_TRAJECTORY", positionGroup);
        h.setGroup("TRAJECTORY_MANAGER", alphaGroup);
        TrajectoryManager::TestFile pos, alp;
        pos.fps = posFps;
        pos.tracks = {{"positionCoM", 3, std::vector<double>(posCoM, posCoM + 3 * nPos)},
                      {"velocityCoM", 3, std::vector<double>(velCoM, velCoM + 3 * nPos)},
                      {"RPY", 3, std::vector<double>(rpyTrack, rpyTrack + 3 * nPos)},
                      {"RPYDot", 3, std::vector<double>(rpyDotTrack, rpyDotTrack + 3 * nPos)}};
        alp.fps = alphaFps;
        alp.tracks = {{"alphaGravity", 1, std::vector<double>(alphaTrack, alphaTrack + nAlpha)}};
        TrajectoryManager::testFiles()["src/trajectories/minimumJerkTrajectory.mat"] = pos;
        TrajectoryManager::testFiles()["src/trajectories/alphaGravity.mat"] = alp;
    }

    QPInput qp;
    qp.setRobot(std::make_shared<Robot>());
    qp.setRobotReference(distinct ? std::make_shared<Robot>() : qp.getRobot());
    auto take = [&](int n) { Eigen::VectorXd v(n); for (int i = 0; i < n; ++i) v(i) = *p++; return v; };
    qp.setThrottleMPC(take(4));
    qp.setThrustDesMPC(take(4));
    qp.setThrustDotDesMPC(take(4));
    qp.setEstimatedThrustDot(take(4));
    qp.setOutputQPJointsPosition(take(Robot::kJoints));

    auto loadTick = [&]() {
        qp.getRobot()->testLoad(p); p += Robot::kBlock;
        if (distinct) { qp.getRobotReference()->testLoad(p); p += Robot::kBlock; }
        qp.setEstimatedThrustDot(take(4));
    };
    loadTick();   // configure-time state
    VariableSamplingMPCGpu mpc;
    const bool latency = argc > 4 && !std::strcmp(argv[3], "latency");
    const bool twoCall = (argc > 3 && !std::strcmp(argv[3], "two-call")) || (argc > 5 && !std::strcmp(argv[5], "two-call"));
    if (twoCall) mpc.setFusedTick(false);
    std::weak_ptr<BipedalLocomotion::ParametersHandler::IParametersHandler> weakHandler = handler;
    if (!configureVariableSamplingMPCGpu(mpc, weakHandler, qp)) {
        std::printf("configure failed: %s\n", mpc.getLastMessage().c_str());
        return dropKey ? 10 : 4;
    }
    if (latency) {
        const int total = std::atoi(argv[4]);
        const double* firstTick = p;
        std::vector<double> us;
        us.reserve(size_t(total));
        Eigen::VectorXd u4(4), q(Robot::kJoints);
        int solved = 0;
        for (int k = 0; k < total + 50; ++k) {
            if (k % nTicks == 0) p = firstTick;   // cycle through the scenario's provider states
            loadTick();
            const auto t0 = std::chrono::steady_clock::now();
            const bool ok = mpc.update(qp);
            mpc.solveMPC();
            const auto t1 = std::chrono::steady_clock::now();
            if (!ok) return 6;
            if (k >= 50) {   // first ticks: code object load, clocks
                us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                solved += mpc.getQPProblemStatus() == VSMPC_STATUS_SOLVED;
            }
            mpc.getThrottleReference(u4); qp.setThrottleMPC(u4);
            mpc.getThrustReference(u4); qp.setThrustDesMPC(u4);
            mpc.getThrustDotReference(u4); qp.setThrustDotDesMPC(u4);
            mpc.getJointsReferencePosition(q); qp.setOutputQPJointsPosition(q);
        }
        std::sort(us.begin(), us.end());
        double mean = 0.0;
        for (double v : us) mean += v;
        std::printf("{\"ticks\": %d, \"solved\": %d, \"p50_us\": %.2f, \"p99_us\": %.2f, \"mean_us\": %.2f, \"form\": \"%s\"}\n", total,
                    solved, us[us.size() / 2], us[size_t(double(us.size()) * 0.99)], mean / double(us.size()),
                    twoCall ? "update: vsmpc_kinematics_batch, solveMPC: vsmpc_solve_batch" : "solveMPC: vsmpc_tick");
        return 0;
    }
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 5;
    const int nIn = mpc.tickMachine().nIn();
    std::vector<double> row(size_t(nIn) + 1 + 12 + Robot::kJoints + 3 + 3 + 1 + 6);
    Eigen::VectorXd t4(4), q23(Robot::kJoints);
    for (int k = 0; k < nTicks; ++k) {
        loadTick();
        if (k % 2 == 0) {   // alternate between the snippet's tick function and the bare calls
            if (!tickVariableSamplingMPCGpu(mpc, qp, q23, t4)) { std::printf("tick %d failed: %d\n", k, mpc.getLastError()); return 6; }
        } else {
            if (!mpc.update(qp)) { std::printf("update failed at tick %d: %d\n", k, mpc.getLastError()); return 6; }
            mpc.solveMPC();
        }
        double* r = row.data();
        for (int i = 0; i < nIn; ++i) *r++ = mpc.tickMachine().record()[i];   // the record that was solved
        *r++ = mpc.getQPProblemStatus();
        if (!mpc.getThrustReference(t4)) return 7;
        for (int i = 0; i < 4; ++i) *r++ = t4(i);
        mpc.getThrustDotReference(t4);
        for (int i = 0; i < 4; ++i) *r++ = t4(i);
        mpc.getThrottleReference(t4);
        for (int i = 0; i < 4; ++i) *r++ = t4(i);
        Eigen::VectorXd wrongSize(5);
        if (mpc.getThrottleReference(wrongSize)) return 8;   // size-checked like the reference
        mpc.getJointsReferencePosition(q23);
        for (int i = 0; i < Robot::kJoints; ++i) *r++ = q23(i);
        for (int i = 0; i < 3; ++i) *r++ = qp.getPosCoMReference()(i);
        for (int i = 0; i < 3; ++i) *r++ = qp.getRPYReference()(i);
        *r++ = qp.getAlphaGravity();
        for (int i = 0; i < 6; ++i) *r++ = qp.getMomentumReference()(i);
        std::fwrite(row.data(), sizeof(double), row.size(), o);
        // harness feedback (src/variable_sampling_mpc.py:124-135)
        mpc.getThrottleReference(t4); qp.setThrottleMPC(t4);
        mpc.getThrustReference(t4); qp.setThrustDesMPC(t4);
        mpc.getThrustDotReference(t4); qp.setThrustDotDesMPC(t4);
        qp.setOutputQPJointsPosition(q23);
    }
    std::fclose(o);
    return 0;
}
