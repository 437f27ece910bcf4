// tests/cpp/refstub/Robot.h -- TEST INFRASTRUCTURE (see README.md): the getters of utils/include/Robot.h:79-298 that the
// MPC path calls, with the reference's return types and cv-qualifiers, backed by arrays a test loads.  No kinematics.
#ifndef REFSTUB_ROBOT_H
#define REFSTUB_ROBOT_H
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Dense>
#include <iDynTree/Direction.h>
#include <iDynTree/EigenHelpers.h>
#include <iDynTree/Transform.h>
#include <iDynTree/Twist.h>
#include <iDynTree/VectorFixSize.h>

class Robot {
public:
    const iDynTree::Twist getBaseVel() const { return m_baseVel; }
    const iDynTree::Transform getBasePose() const { return m_basePose; }
    const size_t getNJoints() const { return m_nJoints; }
    const size_t getNJets() const { return m_nJets; }
    const double getTotalMass() const { return m_totalMass; }
    Eigen::Ref<const Eigen::VectorXd> getJointPos() const { return m_jointPos; }
    std::string getJointName(int jointPos) const { return "joint_" + std::to_string(jointPos); }
    Eigen::Ref<const Eigen::VectorXd> getJetThrusts() const { return m_jetThrusts; }
    const iDynTree::Vector3& getGravity() const { return m_gravity; }
    Eigen::Ref<const Eigen::MatrixXd> getMassMatrix() const { return m_massMatrix; }
    Eigen::Ref<const Eigen::Vector6d> getMomentum(bool inBodyCoord = false) const {
        if (!inBodyCoord) std::abort();   // the path only ever asks for body coordinates
        return m_momentumBody;
    }
    Eigen::Ref<const Eigen::Vector3d> getPositionCoM() const { return m_posCoM; }
    Eigen::Ref<const Eigen::MatrixXd> getJacobianCoM() const { return m_jacobianCoM; }
    const Eigen::MatrixXd getJacobian(const std::string& frameName) {
        for (size_t i = 0; i < m_jetsList.size(); ++i)
            if (m_jetsList[i] == frameName) return m_jacobianJets[i];
        std::abort();
    }
    Eigen::Ref<const Eigen::MatrixXd> getMatrixAmomJets(bool inBodyCoord = false) const {
        if (!inBodyCoord) std::abort();
        return m_AmomBody;
    }
    const std::vector<iDynTree::Direction> getMatrixOfJetAxes() const { return m_jetAxes; }
    const std::vector<Eigen::MatrixXd>& getRelativeJacobianJetsBodyFrame() const { return m_relJacobians; }
    const std::vector<Eigen::Vector3d>& getMatrixOfJetArms() const { return m_jetArms; }
    const std::vector<std::string>& getJetsList() const { return m_jetsList; }

    // ---- test loader: one robot block of the scenario file (tests/fake_provider.py, FakeRobot.block(); row-major arrays)
    static constexpr int kJoints = 23, kJets = 4;
    static constexpr int kBlock = 3 + 3 + 9 + 3 + 6 + 4 + kJoints + 12 + 12 + 4 * 6 * kJoints + 4 * 6 * (6 + kJoints) + 3 * (6 + kJoints) + 36 + 24 + 1 + 3;
    void testLoad(const double* b) {
        iDynTree::Position pos;
        iDynTree::Rotation rot;
        iDynTree::AngVelocity ang;
        for (int i = 0; i < 3; ++i) m_posCoM(i) = *b++;
        for (int i = 0; i < 3; ++i) pos(i) = *b++;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rot(i, j) = *b++;
        for (int i = 0; i < 3; ++i) ang(i) = *b++;
        m_basePose = iDynTree::Transform(rot, pos);
        m_baseVel = iDynTree::Twist(iDynTree::LinVelocity(), ang);
        for (int i = 0; i < 6; ++i) m_momentumBody(i) = *b++;
        m_jetThrusts.resize(kJets);
        for (int i = 0; i < kJets; ++i) m_jetThrusts(i) = *b++;
        m_jointPos.resize(kJoints);
        for (int i = 0; i < kJoints; ++i) m_jointPos(i) = *b++;
        m_jetAxes.assign(kJets, iDynTree::Direction());
        m_jetArms.assign(kJets, Eigen::Vector3d());
        for (auto& a : m_jetAxes)
            for (int k = 0; k < 3; ++k) a(k) = *b++;
        for (auto& a : m_jetArms)
            for (int k = 0; k < 3; ++k) a(k) = *b++;
        m_relJacobians.assign(kJets, Eigen::MatrixXd(6, kJoints));
        for (auto& m : m_relJacobians) b = rowMajor(b, m);
        m_jacobianJets.assign(kJets, Eigen::MatrixXd(6, 6 + kJoints));
        for (auto& m : m_jacobianJets) b = rowMajor(b, m);
        m_jacobianCoM.resize(3, 6 + kJoints);
        b = rowMajor(b, m_jacobianCoM);
        m_massMatrix.resize(6 + kJoints, 6 + kJoints);   // only the base block travels in the scenario (and is read)
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) m_massMatrix(i, j) = *b++;
        m_AmomBody.resize(6, kJets);
        b = rowMajor(b, m_AmomBody);
        m_totalMass = float(*b++);   // utils/include/Robot.h:338 keeps the mass in a float
        for (int i = 0; i < 3; ++i) m_gravity(i) = *b++;
    }

private:
    static const double* rowMajor(const double* b, Eigen::MatrixXd& m) {
        for (Eigen::Index i = 0; i < m.rows(); ++i)
            for (Eigen::Index j = 0; j < m.cols(); ++j) m(i, j) = *b++;
        return b;
    }
    size_t m_nJoints = kJoints, m_nJets = kJets;
    float m_totalMass = 0.0f;
    iDynTree::Twist m_baseVel;
    iDynTree::Transform m_basePose;
    iDynTree::Vector3 m_gravity;
    Eigen::VectorXd m_jointPos, m_jetThrusts;
    Eigen::Vector6d m_momentumBody;
    Eigen::Vector3d m_posCoM;
    Eigen::MatrixXd m_massMatrix, m_jacobianCoM, m_AmomBody;
    std::vector<iDynTree::Direction> m_jetAxes;
    std::vector<Eigen::Vector3d> m_jetArms;
    std::vector<Eigen::MatrixXd> m_relJacobians, m_jacobianJets;
    std::vector<std::string> m_jetsList{"l_arm_jet_turbine", "r_arm_jet_turbine", "chest_l_jet_turbine", "chest_r_jet_turbine"};
};
#endif
