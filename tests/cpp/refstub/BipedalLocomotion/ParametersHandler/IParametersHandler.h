// tests/cpp/refstub/.../IParametersHandler.h -- TEST INFRASTRUCTURE, not BLF (see ../../README.md).
// The overload set of BipedalLocomotion::ParametersHandler::IParametersHandler that the reference's plugins call
// (IMPCProblem.cpp:20-60, costsVSMPC.cpp:28-68, systemDynamicsVSMPC.cpp:17-40,239-272, constraintsVSMPC.cpp:24-45,
// 170-180,294-320): getParameter(name, T&) const -> bool, getGroup(name) -> weak_ptr, setGroup(name, shared_ptr).
#ifndef REFSTUB_BLF_IPARAMETERS_HANDLER_H
#define REFSTUB_BLF_IPARAMETERS_HANDLER_H
#include <memory>
#include <string>
#include <vector>

#include <BipedalLocomotion/GenericContainer/Vector.h>

namespace BipedalLocomotion {
namespace ParametersHandler {

class IParametersHandler {
public:
    using shared_ptr = std::shared_ptr<IParametersHandler>;
    using weak_ptr = std::weak_ptr<IParametersHandler>;

    virtual bool getParameter(const std::string& parameterName, int& parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, double& parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, std::string& parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, bool& parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, std::vector<bool>& parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, GenericContainer::Vector<int>::Ref parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, GenericContainer::Vector<double>::Ref parameter) const = 0;
    virtual bool getParameter(const std::string& parameterName, GenericContainer::Vector<std::string>::Ref parameter) const = 0;
    virtual bool setGroup(const std::string& name, shared_ptr newGroup) = 0;
    virtual weak_ptr getGroup(const std::string& name) const = 0;
    virtual ~IParametersHandler() = default;
};

}  // namespace ParametersHandler
}  // namespace BipedalLocomotion
#endif
