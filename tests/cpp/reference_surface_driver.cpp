// Drives vsmpc_host::VariableSamplingMPCT -- the reference's configure(parametersHandler, qpInput) / update(qpInput) /
// solveMPC() surface (include/VariableSamplingMPC.hpp) -- with FAKE provider classes that expose exactly the members of
// utils/include/Robot.h, utils/include/QPInput.h:12-124 and BLF's IParametersHandler::getParameter the path touches
// (element access through operator(), like Eigen / iDynTree), fed from a scenario file written by the Python test.
//   reference_surface_driver <scenario.bin> <out.bin>
// Scenario (doubles): header[16] | trajectories | config | controlled-joint indices[8] | initial QPInput | per tick
// (configure-time first): robot block (+ reference-robot block when header says the two differ) + estimatedThrustDot[4].
// Output per tick: record[n_in] | status | thrust[4] | thrustDot[4] | throttle[4] | joints[23] | posCoMRef[3] | rpyRef[3] |
// alpha | momentumRef[6].
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "VariableSamplingMPC.hpp"

namespace fake {

struct Vec {   // Eigen::VectorXd / iDynTree::Vector3-like
    std::vector<double> v;
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    int size() const { return int(v.size()); }
};
struct Mat {   // Eigen::MatrixXd / iDynTree::Rotation-like, row-major storage
    int r = 0, c = 0;
    std::vector<double> v;
    double operator()(int i, int j) const { return v[size_t(i) * c + j]; }
};
struct Transform {
    Vec pos; Mat rot;
    const Vec& getPosition() const { return pos; }
    const Mat& getRotation() const { return rot; }
};
struct Twist {
    Vec lin, ang;
    const Vec& getLinearVec3() const { return lin; }
    const Vec& getAngularVec3() const { return ang; }
};

constexpr int NJ = 23, NJETS = 4;
constexpr int ROBOT_BLOCK = 3 + 3 + 9 + 3 + 6 + 4 + NJ + 12 + 12 + 4 * 6 * NJ + 4 * 6 * (6 + NJ) + 3 * (6 + NJ) + 36 + 24 + 1 + 3;

class Robot {   // the getters of utils/include/Robot.h the path uses
public:
    void load(const double* b) {
        auto take = [&](int n) { std::vector<double> o(b, b + n); b += n; return o; };
        com.v = take(3); pose.pos.v = take(3);
        pose.rot.r = pose.rot.c = 3; pose.rot.v = take(9);
        vel.ang.v = take(3); vel.lin.v = {0, 0, 0};
        mom.v = take(6); thrust.v = take(4); q.v = take(NJ);
        axes.assign(NJETS, Vec{}); arms.assign(NJETS, Vec{});
        for (auto& a : axes) a.v = take(3);
        for (auto& a : arms) a.v = take(3);
        rel.assign(NJETS, Mat{});
        for (auto& m : rel) { m.r = 6; m.c = NJ; m.v = take(6 * NJ); }
        jac.assign(NJETS, Mat{});
        for (auto& m : jac) { m.r = 6; m.c = 6 + NJ; m.v = take(6 * (6 + NJ)); }
        jcom.r = 3; jcom.c = 6 + NJ; jcom.v = take(3 * (6 + NJ));
        M.r = 6; M.c = 6; M.v = take(36);   // (only the base block is ever read)
        amom.r = 6; amom.c = 4; amom.v = take(24);
        mass = *b++; grav.v = take(3);
    }
    size_t getNJoints() const { return NJ; }
    size_t getNJets() const { return NJETS; }
    double getTotalMass() const { return mass; }
    Transform getBasePose() const { return pose; }
    Twist getBaseVel() const { return vel; }
    const Vec& getPositionCoM() const { return com; }
    const Vec& getMomentum(bool inBodyCoord = false) const { if (!inBodyCoord) std::abort(); return mom; }
    const Vec& getGravity() const { return grav; }
    const Vec& getJetThrusts() const { return thrust; }
    const Vec& getJointPos() const { return q; }
    const Mat& getMassMatrix() const { return M; }
    const Mat& getMatrixAmomJets(bool inBodyCoord = false) const { if (!inBodyCoord) std::abort(); return amom; }
    std::vector<Vec> getMatrixOfJetAxes() const { return axes; }
    const std::vector<Vec>& getMatrixOfJetArms() const { return arms; }
    const std::vector<Mat>& getRelativeJacobianJetsBodyFrame() const { return rel; }
    const std::vector<std::string>& getJetsList() const { return jets; }
    Mat getJacobian(const std::string& frameName) {
        for (int i = 0; i < NJETS; ++i) if (jets[i] == frameName) return jac[i];
        std::abort();
    }
    const Mat& getJacobianCoM() const { return jcom; }
    std::string getJointName(int i) const { return "joint_" + std::to_string(i); }

private:
    Vec com, mom, thrust, q, grav;
    Transform pose; Twist vel;
    std::vector<Vec> axes, arms;
    std::vector<Mat> rel, jac;
    Mat jcom, M, amom;
    double mass = 0.0;
    std::vector<std::string> jets{"l_arm_jet_turbine", "r_arm_jet_turbine", "chest_l_jet_turbine", "chest_r_jet_turbine"};
};

class QPInput {   // utils/include/QPInput.h: the members the path touches
public:
    std::shared_ptr<Robot> getRobot() const { return robot; }
    std::shared_ptr<Robot> getRobotReference() const { return robotReference; }
    const Vec& getThrottleMPC() const { return throttleMPC; }
    const Vec& getThrustDesMPC() const { return thrustDesMPC; }
    const Vec& getThrustDotDesMPC() const { return thrustDotDesMPC; }
    const Vec& getEstimatedThrustDot() const { return estimatedThrustDot; }
    const Vec& getOutputQPJointsPosition() const { return outputQPJointsPosition; }
    const Vec& getPosCoMReference() const { return posCoMReference; }
    const Vec& getRPYReference() const { return rpyReference; }
    void setPosCoMReference(const double* v) { posCoMReference.v.assign(v, v + 3); }
    void setRPYReference(const double* v) { rpyReference.v.assign(v, v + 3); }
    void setMomentumReference(const double* v) { momentumReference.v.assign(v, v + 6); }
    void setAlphaGravity(double a) { alphaGravity = a; }

    std::shared_ptr<Robot> robot, robotReference;
    Vec throttleMPC{{0, 0, 0, 0}}, thrustDesMPC{{0, 0, 0, 0}}, thrustDotDesMPC{{0, 0, 0, 0}}, estimatedThrustDot{{0, 0, 0, 0}};
    Vec outputQPJointsPosition{std::vector<double>(NJ, 0.0)};
    Vec posCoMReference{{0, 0, 0}}, rpyReference{{0, 0, 0}}, momentumReference{std::vector<double>(6, 0.0)};
    double alphaGravity = 0.0;
};

class ParametersHandler {   // BLF IParametersHandler::getParameter(name, value) -> bool
public:
    std::map<std::string, double> scalars;
    std::map<std::string, std::vector<double>> vectors;
    std::map<std::string, std::string> strings;
    std::map<std::string, std::vector<std::string>> stringLists;
    bool getParameter(const std::string& k, int& v) const { auto it = scalars.find(k); if (it == scalars.end()) return false; v = int(it->second); return true; }
    bool getParameter(const std::string& k, double& v) const { auto it = scalars.find(k); if (it == scalars.end()) return false; v = it->second; return true; }
    bool getParameter(const std::string& k, bool& v) const { auto it = scalars.find(k); if (it == scalars.end()) return false; v = it->second != 0.0; return true; }
    bool getParameter(const std::string& k, std::string& v) const { auto it = strings.find(k); if (it == strings.end()) return false; v = it->second; return true; }
    bool getParameter(const std::string& k, std::vector<double>& v) const { auto it = vectors.find(k); if (it == vectors.end()) return false; v = it->second; return true; }
    bool getParameter(const std::string& k, std::vector<std::string>& v) const { auto it = stringLists.find(k); if (it == stringLists.end()) return false; v = it->second; return true; }
};

}  // namespace fake

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
    fake::ParametersHandler h;
    const char* scalarKeys[] = {"nIter", "nIterSmall", "controlHorizon", "useJetDynamic", "periodMPC", "periodMPCSmallSteps",
                                "periodMPCLargeSteps"};
    for (const char* k : scalarKeys) h.scalars[k] = *p++;
    const char* vecKeys[] = {"weightCoMPos", "weightCoMPosError", "weightLinMom", "weightRPY", "weightRPYError", "weightAngMom"};
    for (const char* k : vecKeys) { h.vectors[k] = std::vector<double>(p, p + 3); p += 3; }
    h.vectors["weightDeltaJoint"] = std::vector<double>(p, p + 8); p += 8;
    const char* tailKeys[] = {"weightThrottle", "weightInitialThrottle", "weightRegularizationJointPos", "throttleMin", "throttleMax"};
    for (const char* k : tailKeys) h.scalars[k] = *p++;
    h.scalars["useEstimatedThrust"] = useEst ? 1.0 : 0.0;
    h.strings["jointsLambdaOption"] = constantLambda ? "constant" : "unfiltered";
    std::vector<std::string> names;
    for (int i = 0; i < 8; ++i) names.push_back("joint_" + std::to_string(int(*p++)));
    h.stringLists["controlledJoints"] = names;
    if (dropKey) h.scalars.erase("weightThrottle");

    fake::QPInput qp;
    qp.robot = std::make_shared<fake::Robot>();
    qp.robotReference = distinct ? std::make_shared<fake::Robot>() : qp.robot;
    for (int i = 0; i < 4; ++i) qp.throttleMPC(i) = *p++;
    for (int i = 0; i < 4; ++i) qp.thrustDesMPC(i) = *p++;
    for (int i = 0; i < 4; ++i) qp.thrustDotDesMPC(i) = *p++;
    for (int i = 0; i < 4; ++i) qp.estimatedThrustDot(i) = *p++;
    for (int i = 0; i < fake::NJ; ++i) qp.outputQPJointsPosition(i) = *p++;

    auto position = std::make_shared<vsmpc_host::Trajectory>();
    auto alpha = std::make_shared<vsmpc_host::Trajectory>();
    const int desPos = int(1.0 / h.scalars["periodMPCLargeSteps"]);   // costsVSMPC.cpp:68
    const int desAlpha = int(1.0 / h.scalars["periodMPC"]);            // systemDynamicsVSMPC.cpp:272
    position->add("positionCoM", posCoM, nPos, 3, posFps, desPos);
    position->add("velocityCoM", velCoM, nPos, 3, posFps, desPos);
    position->add("RPY", rpyTrack, nPos, 3, posFps, desPos);
    position->add("RPYDot", rpyDotTrack, nPos, 3, posFps, desPos);
    alpha->add("alphaGravity", alphaTrack, nAlpha, 1, alphaFps, desAlpha);

    auto loadTick = [&]() {
        qp.robot->load(p); p += fake::ROBOT_BLOCK;
        if (distinct) { qp.robotReference->load(p); p += fake::ROBOT_BLOCK; }
        for (int i = 0; i < 4; ++i) qp.estimatedThrustDot(i) = *p++;
    };
    loadTick();   // configure-time state
    vsmpc_host::VariableSamplingMPCT<fake::QPInput> mpc;
    mpc.setTrajectories(position, alpha);
    if (!mpc.configure(h, qp)) {
        std::printf("configure failed: %s\n", mpc.getLastMessage().c_str());
        return dropKey ? 10 : 4;
    }
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 5;
    const int nIn = mpc.tickMachine().nIn();
    std::vector<double> row(size_t(nIn) + 1 + 12 + fake::NJ + 3 + 3 + 1 + 6);
    for (int k = 0; k < nTicks; ++k) {
        loadTick();
        if (!mpc.update(qp)) { std::printf("update failed at tick %d: %d\n", k, mpc.getLastError()); return 6; }
        mpc.solveMPC();
        double* r = row.data();
        for (int i = 0; i < nIn; ++i) *r++ = mpc.tickMachine().record()[i];
        *r++ = mpc.getQPProblemStatus();
        fake::Vec t4{std::vector<double>(4)}, q23{std::vector<double>(fake::NJ)};
        if (!mpc.getThrustReference(t4)) return 7;
        for (int i = 0; i < 4; ++i) *r++ = t4(i);
        mpc.getThrustDotReference(t4);
        for (int i = 0; i < 4; ++i) *r++ = t4(i);
        mpc.getThrottleReference(t4);
        for (int i = 0; i < 4; ++i) *r++ = t4(i);
        fake::Vec wrongSize{std::vector<double>(5)};
        if (mpc.getThrottleReference(wrongSize)) return 8;   // size-checked like the reference
        mpc.getJointsReferencePosition(q23);
        for (int i = 0; i < fake::NJ; ++i) *r++ = q23(i);
        for (int i = 0; i < 3; ++i) *r++ = qp.posCoMReference(i);
        for (int i = 0; i < 3; ++i) *r++ = qp.rpyReference(i);
        *r++ = qp.alphaGravity;
        for (int i = 0; i < 6; ++i) *r++ = qp.momentumReference(i);
        std::fwrite(row.data(), sizeof(double), row.size(), o);
        // harness feedback (src/variable_sampling_mpc.py:124-135)
        mpc.getThrottleReference(qp.throttleMPC);
        mpc.getThrustReference(qp.thrustDesMPC);
        mpc.getThrustDotReference(qp.thrustDotDesMPC);
        mpc.getJointsReferencePosition(qp.outputQPJointsPosition);
    }
    std::fclose(o);
    return 0;
}
