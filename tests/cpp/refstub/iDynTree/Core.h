// tests/cpp/refstub/iDynTree/Core.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md).
// Element access of the iDynTree value types utils/include/Robot.h returns on the path: VectorFixSize<3>::operator()(i),
// Rotation::operator()(r, c) (row, column), Transform::getPosition / getRotation, Twist::getLinearVec3 / getAngularVec3.
#ifndef REFSTUB_IDYNTREE_CORE_H
#define REFSTUB_IDYNTREE_CORE_H
#include <cstddef>
#include <Eigen/Core>

namespace iDynTree {
using FrameIndex = std::ptrdiff_t;

class Vector3 {
public:
    double operator()(const unsigned int index) const { return m_data[index]; }
    double& operator()(const unsigned int index) { return m_data[index]; }
    const double* data() const { return m_data; }
    double* data() { return m_data; }
    size_t size() const { return 3; }

private:
    double m_data[3] = {0.0, 0.0, 0.0};
};
class Position : public Vector3 {};
class Direction : public Vector3 {};
class LinVelocity : public Vector3 {};
class AngVelocity : public Vector3 {};

class Rotation {
public:
    double operator()(const unsigned int row, const unsigned int col) const { return m_data[3 * row + col]; }
    double& operator()(const unsigned int row, const unsigned int col) { return m_data[3 * row + col]; }

private:
    double m_data[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
};

class Transform {
public:
    Transform() = default;
    Transform(const Rotation& rot, const Position& origin) : m_pos(origin), m_rot(rot) {}
    const Position& getPosition() const { return m_pos; }
    const Rotation& getRotation() const { return m_rot; }

private:
    Position m_pos;
    Rotation m_rot;
};

class Twist {
public:
    Twist() = default;
    Twist(const LinVelocity& lin, const AngVelocity& ang) : m_lin(lin), m_ang(ang) {}
    const LinVelocity& getLinearVec3() const { return m_lin; }
    const AngVelocity& getAngularVec3() const { return m_ang; }

private:
    LinVelocity m_lin;
    AngVelocity m_ang;
};
}  // namespace iDynTree

// iDynTree/EigenHelpers.h is where the reference's `Eigen::Vector6d` / `Eigen::Matrix6d` come from
namespace Eigen {
using Vector6d = Matrix<double, 6, 1>;
using Matrix6d = Matrix<double, 6, 6>;
}  // namespace Eigen
#endif
