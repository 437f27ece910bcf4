// pybind11 shim `bindingsMPC` with the reference's Python surface
// (momentum-based-linear-mpc-lib/bindings/python/MPCPyBindings.cpp:12-91): class VariableSamplingMPC with
// configure / update / solveMPC / get*Reference / getFinal* / getNStatesMPC / getNInputMPC.
// Differences forced by the boundary: configure() takes the VS_MPC_CONFIG keys as a dict instead of
// a BLF IParametersHandler + QPInput, update() takes the per-tick input record (VSMPC_IN_* layout) instead of a
// QPInput backed by a live iDynTree Robot.  Host-only glue over include/VariableSamplingMPC.hpp -> libvsmpc.so.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/VariableSamplingMPC.hpp"

namespace py = pybind11;
using vsmpc_host::VariableSamplingMPC;

namespace {

template <class T>
T get(const py::dict& d, const char* k) {
    if (!d.contains(k)) throw std::invalid_argument(std::string("Parameter '") + k + "' not found in the config");
    return d[k].cast<T>();
}
void get_vec(const py::dict& d, const char* k, double* out, size_t n) {
    auto v = get<std::vector<double>>(d, k);
    if (v.size() != n) throw std::invalid_argument(std::string("Parameter '") + k + "' has the wrong size");
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
}

vsmpc_config config_from_dict(const py::dict& d) {  // keys of src/config/vs_mcp_config.xml:7-43
    vsmpc_config c{};
    c.n_iter = get<int>(d, "nIter");
    c.n_iter_small = get<int>(d, "nIterSmall");
    c.control_horizon = get<int>(d, "controlHorizon");
    c.use_jet_dynamic = get<bool>(d, "useJetDynamic") ? 1 : 0;
    c.period_mpc = get<double>(d, "periodMPC");
    c.period_small = get<double>(d, "periodMPCSmallSteps");
    c.period_large = get<double>(d, "periodMPCLargeSteps");
    get_vec(d, "weightCoMPos", c.w_com_pos, 3);
    get_vec(d, "weightCoMPosError", c.w_com_pos_err, 3);
    get_vec(d, "weightLinMom", c.w_lin_mom, 3);
    get_vec(d, "weightRPY", c.w_rpy, 3);
    get_vec(d, "weightRPYError", c.w_rpy_err, 3);
    get_vec(d, "weightAngMom", c.w_ang_mom, 3);
    get_vec(d, "weightDeltaJoint", c.w_delta_joint, 8);
    c.w_throttle = get<double>(d, "weightThrottle");
    c.w_initial_throttle = get<double>(d, "weightInitialThrottle");
    c.w_reg_joint_pos = get<double>(d, "weightRegularizationJointPos");
    c.throttle_min = get<double>(d, "throttleMin");
    c.throttle_max = get<double>(d, "throttleMax");
    return c;
}

py::array_t<double> vec(const std::vector<double>& v) { return py::array_t<double>(v.size(), v.data()); }

}  // namespace

PYBIND11_MODULE(bindingsMPC, m) {
    m.doc() = "MI355X-backed drop-in for momentum_based_mpc.bindingsMPC (VariableSamplingMPC only)";
    py::class_<VariableSamplingMPC>(m, "VariableSamplingMPC")
        .def(py::init<>())
        .def("configure",
             [](VariableSamplingMPC& self, const py::dict& params, py::object jointPos, py::object rpy0, int device) {
                 const vsmpc_config c = config_from_dict(params);
                 std::vector<double> q(vsmpc_host::kRobotJoints, 0.0), r(3, 0.0);
                 if (!jointPos.is_none()) q = jointPos.cast<std::vector<double>>();
                 if (!rpy0.is_none()) r = rpy0.cast<std::vector<double>>();
                 if (q.size() != size_t(vsmpc_host::kRobotJoints) || r.size() != 3) return false;
                 return self.configure(c, q.data(), r.data(), device);
             },
             py::arg("parametersHandler"), py::arg("jointPositions") = py::none(), py::arg("initialRPY") = py::none(),
             py::arg("device") = 0)
        .def("update",
             [](VariableSamplingMPC& self, py::array_t<double, py::array::c_style | py::array::forcecast> rec,
                bool applyTickState) {
                 if (rec.size() != self.inputDoubles()) return false;
                 return self.update(rec.data(), applyTickState);
             },
             py::arg("mpcInput"), py::arg("applyTickState") = false)
        .def("solveMPC", &VariableSamplingMPC::solveMPC)
        .def("getQPProblemStatus", &VariableSamplingMPC::getQPProblemStatus)
        .def("getMPCSolution", [](VariableSamplingMPC& self) {
            const auto& x = self.getSolution();
            const size_t off = size_t(VSMPC_N_STATES) * (self.config().n_iter + 1);
            return py::array_t<double>(x.size() - off, x.data() + off);
        })
        .def("getJointsReferencePosition", [](VariableSamplingMPC& self) {   // MPCPyBindings.cpp:40-46 (23 joints)
            std::vector<double> v(vsmpc_host::kRobotJoints);
            self.getJointsReferencePosition(v.data(), int(v.size()));
            return vec(v);
        })
        .def("getThrottleReference", [](VariableSamplingMPC& self) { std::vector<double> v(4); self.getThrottleReference(v.data(), 4); return vec(v); })
        .def("getThrustReference", [](VariableSamplingMPC& self) { std::vector<double> v(4); self.getThrustReference(v.data(), 4); return vec(v); })
        .def("getThrustDotReference", [](VariableSamplingMPC& self) { std::vector<double> v(4); self.getThrustDotReference(v.data(), 4); return vec(v); })
        .def("getFinalCoMPosition", [](VariableSamplingMPC& self) { std::vector<double> v(3); self.getFinalCoMPosition(v.data(), 3); return vec(v); })
        .def("getFinalLinMom", [](VariableSamplingMPC& self) { std::vector<double> v(3); self.getFinalLinMom(v.data(), 3); return vec(v); })
        .def("getFinalRPY", [](VariableSamplingMPC& self) { std::vector<double> v(3); self.getFinalRPY(v.data(), 3); return vec(v); })
        .def("getFinalAngMom", [](VariableSamplingMPC& self) { std::vector<double> v(3); self.getFinalAngMom(v.data(), 3); return vec(v); })
        .def("getNStatesMPC", &VariableSamplingMPC::getNStatesMPC)
        .def("getNInputMPC", &VariableSamplingMPC::getNInputMPC);
}
