// momentum-based-linear-mpc-lib/src/variableSamplingMPC/variableSamplingMPCGpu.cpp -- the reference-side binding.
// Build inside the reference: add this repo's include/ to the include path and link libvsmpc.so.
#include <VariableSamplingMPC.hpp>          // this repo's include/ (pulls vsmpc.h)
#include <QPInput.h>                        // utils/include/QPInput.h
#include <TrajectoryManager.h>              // utils/include/TrajectoryManager.h
#include <BipedalLocomotion/ParametersHandler/YarpImplementation.h>

using VariableSamplingMPCGpu = vsmpc_host::VariableSamplingMPCT<QPInput, TrajectoryManager>;

// The two TrajectoryManager objects the reference's plugins own (costsVSMPC.cpp:47-68, systemDynamicsVSMPC.cpp:263-272),
// configured from the same groups, at the same rates.
inline bool configureVariableSamplingMPCGpu(
    VariableSamplingMPCGpu& mpc,
    std::weak_ptr<BipedalLocomotion::ParametersHandler::IParametersHandler> parametersHandler,
    QPInput& qpInput)
{
    using BipedalLocomotion::ParametersHandler::YarpImplementation;
    auto ptr = parametersHandler.lock();
    double periodMPC = 0.0, periodMPCLargeSteps = 0.0;
    if (!ptr || !ptr->getParameter("periodMPC", periodMPC)
        || !ptr->getParameter("periodMPCLargeSteps", periodMPCLargeSteps))
        return false;
    auto position = std::make_shared<TrajectoryManager>();
    auto alpha = std::make_shared<TrajectoryManager>();
    auto positionGroup = std::make_shared<YarpImplementation>();
    auto alphaGroup = std::make_shared<YarpImplementation>();
    positionGroup->setGroup("TRAJECTORY_MANAGER", ptr->getGroup("POSITION_TRAJECTORY").lock());
    alphaGroup->setGroup("TRAJECTORY_MANAGER", ptr->getGroup("TRAJECTORY_MANAGER").lock());
    if (!position->configure(positionGroup, 1 / periodMPCLargeSteps) || !alpha->configure(alphaGroup, 1 / periodMPC))
        return false;
    mpc.setTrajectories(position, alpha);
    return mpc.configure(parametersHandler, qpInput);   // IMPCProblem::configure(weak_ptr, QPInput&), IMPCProblem.h:35
}

// Per tick, exactly as src/variable_sampling_mpc.py:111-135 drives the reference's class:
inline bool tickVariableSamplingMPCGpu(VariableSamplingMPCGpu& mpc, QPInput& qpInput,
                                       Eigen::Ref<Eigen::VectorXd> jointsReferencePosition,   // 23
                                       Eigen::Ref<Eigen::VectorXd> throttleReference)         // 4
{
    if (!mpc.update(qpInput)) return false;
    mpc.solveMPC();
    if (mpc.getQPProblemStatus() != VSMPC_STATUS_SOLVED) return false;   // OsqpEigen::Status::Solved
    return mpc.getJointsReferencePosition(jointsReferencePosition) && mpc.getThrottleReference(throttleReference);
}
