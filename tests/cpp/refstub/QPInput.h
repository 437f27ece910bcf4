// tests/cpp/refstub/QPInput.h -- TEST INFRASTRUCTURE (see README.md): the members of utils/include/QPInput.h:12-124 the
// MPC path calls, parameter and return types as the reference declares them (setters :41,45,67,75).
#ifndef REFSTUB_QPINPUT_H
#define REFSTUB_QPINPUT_H
#include <atomic>
#include <memory>

#include "Robot.h"
#include <BipedalLocomotion/YarpUtilities/VectorsCollectionServer.h>

class QPInput {
public:
    const std::shared_ptr<Robot> getRobot() const { return m_robot; }
    void setRobot(std::shared_ptr<Robot> robot) { m_robot = robot; }
    const std::shared_ptr<Robot> getRobotReference() const { return m_robotReference; }
    void setRobotReference(std::shared_ptr<Robot> robotReference) { m_robotReference = robotReference; }

    const Eigen::Ref<const Eigen::VectorXd> getOutputQPJointsPosition() const { return m_outputQPJointsPosition; }
    void setOutputQPJointsPosition(const Eigen::Ref<const Eigen::VectorXd> outputQPJointsPosition) { m_outputQPJointsPosition = outputQPJointsPosition; }
    const Eigen::Ref<const Eigen::Vector3d> getPosCoMReference() { return m_posCoMReference; }
    void setPosCoMReference(const Eigen::Ref<const Eigen::Vector3d> posCoMReference) { m_posCoMReference = posCoMReference; }
    const Eigen::Ref<const Eigen::Vector3d> getRPYReference() const { return m_RPYReference; }
    void setRPYReference(const Eigen::Ref<const Eigen::Vector3d> RPYReference) { m_RPYReference = RPYReference; }
    const Eigen::Vector6d& getMomentumReference() const { return m_momentumReference; }
    void setMomentumReference(const Eigen::Vector6d& momentumReference) { m_momentumReference = momentumReference; }
    void setAlphaGravity(const double alphaGravity) { m_alphaGravity = alphaGravity; }
    const double getAlphaGravity() const { return m_alphaGravity; }

    const Eigen::Ref<const Eigen::VectorXd> getThrustDesMPC() const { return m_thrustDesMPC; }
    void setThrustDesMPC(const Eigen::Ref<const Eigen::VectorXd> thrustDesMPC) { m_thrustDesMPC = thrustDesMPC; }
    const Eigen::Ref<const Eigen::VectorXd> getThrustDotDesMPC() const { return m_thrustDotDesMPC; }
    void setThrustDotDesMPC(const Eigen::Ref<const Eigen::VectorXd> thrustDotDesMPC) { m_thrustDotDesMPC = thrustDotDesMPC; }
    const Eigen::Ref<const Eigen::VectorXd> getEstimatedThrustDot() const { return m_estimatedThrustDot; }
    void setEstimatedThrustDot(const Eigen::Ref<const Eigen::VectorXd> estimatedThrustDot) { m_estimatedThrustDot = estimatedThrustDot; }
    const Eigen::Ref<const Eigen::VectorXd> getThrottleMPC() const { return m_throttleMPC; }
    void setThrottleMPC(const Eigen::Ref<const Eigen::VectorXd> throttleMPC) { m_throttleMPC = throttleMPC; }

private:
    std::shared_ptr<Robot> m_robot, m_robotReference;
    Eigen::Vector3d m_posCoMReference, m_RPYReference;
    Eigen::VectorXd m_outputQPJointsPosition, m_thrustDesMPC, m_thrustDotDesMPC, m_throttleMPC, m_estimatedThrustDot;
    Eigen::Vector6d m_momentumReference;
    std::atomic<double> m_alphaGravity{0.0};
};
#endif
