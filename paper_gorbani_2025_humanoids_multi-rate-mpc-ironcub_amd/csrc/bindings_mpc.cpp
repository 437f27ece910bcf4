// pybind11 shim `bindingsMPC` with the reference's Python surface
// (momentum-based-linear-mpc-lib/bindings/python/MPCPyBindings.cpp:12-91): class VariableSamplingMPC with
//     configure(parametersHandler, mpcInput) -> bool        (:24-32)
//     update(mpcInput) -> bool                              (:37)
//     solveMPC / getMPCSolution / get*Reference / getFinal* / getNStatesMPC / getNInputMPC   (:38-90)
// `parametersHandler` and `mpcInput` are Python objects: the handler may be BLF's Python handler
// (get_parameter_int / _float / _bool / _string / _vector_float / _vector_string, get_group -- what
// src/variable_sampling_mpc.py:37-40 holds), an object with getParameter(name), or a mapping; `mpcInput` any object with
// the QPInput getters / setters the path uses (utils/include/QPInput.h:12-124) whose getRobot() / getRobotReference()
// return objects with the Robot getters of utils/include/Robot.h (arrays come back as numpy).  Both are read through the
// SAME templates (include/VariableSamplingMPC.hpp: readRobot / readQPInput / readParameters -> TickMachine) that compile
// against the reference's C++ classes; this file only adapts Python attribute access to those member names.
// The trajectories the reference reads from MAT files (groups POSITION_TRAJECTORY / TRAJECTORY_MANAGER,
// src/config/vs_mcp_config.xml:34-40: a `trajectoryFile` name each, loaded by TrajectoryManager.cpp:40-140) reach this
// module either as arrays under those group names (positionCoM, velocityCoM, RPY, RPYDot, fps | alphaGravity, fps) or --
// the handler exactly as the harness read it from XML -- as the `trajectoryFile` name, which is handed to a LOADER
// callable (setTrajectoryLoader / the module-level set_trajectory_loader; trajectory_io.load_mat_trajectory is the
// default the package installs): loader(fileName) -> mapping {variable: array, "fps": int}.  MAT decoding itself stays
// in Python.
// A second, record-level entry -- update(record) after configureRecord(dict, jointPositions, initialRPY) -- serves callers
// that already hold the vsmpc_input record (its own name: `configure` has exactly the reference's signature, so that
// configure(handler, mpcInput, device) with a dict handler cannot be mis-dispatched).  Host-only glue -> libvsmpc.so; no numerics here.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/VariableSamplingMPC.hpp"

namespace py = pybind11;
using vsmpc_host::Trajectory;

namespace {

using Arr = py::array_t<double, py::array::c_style | py::array::forcecast>;

struct PyVec {   // Eigen::VectorXd-like view of whatever a Python getter returned
    Arr a;
    PyVec() = default;
    explicit PyVec(const py::object& o) : a(Arr::ensure(o)) { if (!a) throw std::invalid_argument("expected an array"); }
    double operator()(int i) const { return a.data()[i]; }
    int size() const { return int(a.size()); }
};
struct PyMat {
    Arr a;
    PyMat() = default;
    explicit PyMat(const py::object& o) : a(Arr::ensure(o)) { if (!a || a.ndim() != 2) throw std::invalid_argument("expected a 2-D array"); }
    double operator()(int i, int j) const { return a.data()[size_t(i) * a.shape(1) + j]; }
};
struct PyTransform {
    PyVec pos; PyMat rot;
    const PyVec& getPosition() const { return pos; }
    const PyMat& getRotation() const { return rot; }
};
struct PyTwist {
    PyVec ang;
    const PyVec& getAngularVec3() const { return ang; }
};

py::object rotation_from_rpy(const py::object& rpy) {   // Rz(y) Ry(p) Rx(r), iDynTree::Rotation::RPY
    const PyVec v(rpy);
    const double cr = std::cos(v(0)), sr = std::sin(v(0)), cp = std::cos(v(1)), sp = std::sin(v(1)), cy = std::cos(v(2)), sy = std::sin(v(2));
    Arr R({3, 3});
    double* m = R.mutable_data();
    m[0] = cy * cp; m[1] = cy * sp * sr - sy * cr; m[2] = cy * sp * cr + sy * sr;
    m[3] = sy * cp; m[4] = sy * sp * sr + cy * cr; m[5] = sy * sp * cr - cy * sr;
    m[6] = -sp; m[7] = cp * sr; m[8] = cp * cr;
    return std::move(R);
}

// utils/include/Robot.h member names over a Python object
class PyRobot {
public:
    explicit PyRobot(py::object o) : m(std::move(o)) {}
    size_t getNJoints() const { return py::hasattr(m, "getNJoints") ? m.attr("getNJoints")().cast<size_t>() : size_t(PyVec(m.attr("getJointPos")()).size()); }
    size_t getNJets() const { return py::hasattr(m, "getNJets") ? m.attr("getNJets")().cast<size_t>() : size_t(PyVec(m.attr("getJetThrusts")()).size()); }
    double getTotalMass() const { return m.attr("getTotalMass")().cast<double>(); }
    PyTransform getBasePose() const {
        PyTransform t;
        t.pos = PyVec(m.attr("getBasePosition")());
        // the reference's Python Robot offers RPY (flightCtrlPyBindings.cpp:75-78); a provider may offer the matrix
        t.rot = PyMat(py::hasattr(m, "getBaseRotation") ? py::object(m.attr("getBaseRotation")())
                                                        : rotation_from_rpy(m.attr("getBaseOrientation")()));
        return t;
    }
    PyTwist getBaseVel() const { return PyTwist{PyVec(m.attr("getBaseAngVel")())}; }
    PyVec getPositionCoM() const { return PyVec(m.attr("getPositionCoM")()); }
    PyVec getMomentum(bool inBodyCoord) const { return PyVec(m.attr("getMomentum")(inBodyCoord)); }
    PyVec getGravity() const { return PyVec(m.attr("getGravity")()); }
    PyVec getJetThrusts() const { return PyVec(m.attr("getJetThrusts")()); }
    PyVec getJointPos() const { return PyVec(m.attr("getJointPos")()); }
    PyMat getMassMatrix() const { return PyMat(m.attr("getMassMatrix")()); }
    PyMat getMatrixAmomJets(bool inBodyCoord) const { return PyMat(m.attr("getMatrixAmomJets")(inBodyCoord)); }
    std::vector<PyVec> getMatrixOfJetAxes() const { return rows(m.attr("getMatrixOfJetAxes")()); }
    std::vector<PyVec> getMatrixOfJetArms() const { return rows(m.attr("getMatrixOfJetArms")()); }
    std::vector<PyMat> getRelativeJacobianJetsBodyFrame() const {
        std::vector<PyMat> out;
        for (py::handle h : m.attr("getRelativeJacobianJetsBodyFrame")()) out.emplace_back(py::reinterpret_borrow<py::object>(h));
        return out;
    }
    std::vector<std::string> getJetsList() const { return m.attr("getJetsList")().cast<std::vector<std::string>>(); }
    PyMat getJacobian(const std::string& frameName) const { return PyMat(m.attr("getJacobian")(frameName)); }
    PyMat getJacobianCoM() const { return PyMat(m.attr("getJacobianCoM")()); }
    std::string getJointName(int i) const {
        if (py::hasattr(m, "getJointName")) return m.attr("getJointName")(i).cast<std::string>();
        if (py::hasattr(m, "getAxesList")) return m.attr("getAxesList")().cast<std::vector<std::string>>().at(i);
        return "joint_" + std::to_string(i);
    }

private:
    static std::vector<PyVec> rows(const py::object& o) {
        std::vector<PyVec> out;
        for (py::handle h : o) out.emplace_back(py::reinterpret_borrow<py::object>(h));
        return out;
    }
    py::object m;
};

// utils/include/QPInput.h member names over a Python object
class PyQPInput {
public:
    explicit PyQPInput(py::object o) : m(std::move(o)) {}
    std::unique_ptr<PyRobot> getRobot() const { return std::make_unique<PyRobot>(m.attr("getRobot")()); }
    std::unique_ptr<PyRobot> getRobotReference() const { return std::make_unique<PyRobot>(m.attr("getRobotReference")()); }
    PyVec getThrottleMPC() const { return PyVec(m.attr("getThrottleMPC")()); }
    PyVec getThrustDesMPC() const { return PyVec(m.attr("getThrustDesMPC")()); }
    PyVec getThrustDotDesMPC() const { return PyVec(m.attr("getThrustDotDesMPC")()); }
    PyVec getEstimatedThrustDot() const { return PyVec(m.attr("getEstimatedThrustDot")()); }
    PyVec getOutputQPJointsPosition() const { return PyVec(m.attr("getOutputQPJointsPosition")()); }
    PyVec getPosCoMReference() const { return PyVec(m.attr("getPosCoMReference")()); }
    PyVec getRPYReference() const { return PyVec(m.attr("getRPYReference")()); }
    // (vsmpc_host::fixedArg<N> hands over an Eigen::Map when Eigen is installed, a FixedView otherwise: anything with data())
    template <class V> void setPosCoMReference(const V& v) { set("setPosCoMReference", v.data(), 3); }
    template <class V> void setRPYReference(const V& v) { set("setRPYReference", v.data(), 3); }
    template <class V> void setMomentumReference(const V& v) { set("setMomentumReference", v.data(), 6); }
    void setAlphaGravity(double a) { if (py::hasattr(m, "setAlphaGravity")) m.attr("setAlphaGravity")(a); }

private:
    void set(const char* name, const double* v, int n) {
        if (py::hasattr(m, name)) m.attr(name)(Arr(n, v));   // (a binding without the setter keeps the value inside the MPC only)
    }
    py::object m;
};

// BLF IParametersHandler::getParameter(name, value) -> bool over a Python handler / mapping
class PyParams {
public:
    explicit PyParams(py::object o) : m(std::move(o)) {}
    bool getParameter(const std::string& k, int& v) const { return fetch(k, "get_parameter_int", v); }
    bool getParameter(const std::string& k, double& v) const { return fetch(k, "get_parameter_float", v); }
    bool getParameter(const std::string& k, bool& v) const { return fetch(k, "get_parameter_bool", v); }
    bool getParameter(const std::string& k, std::string& v) const { return fetch(k, "get_parameter_string", v); }
    bool getParameter(const std::string& k, std::vector<double>& v) const { return fetch(k, "get_parameter_vector_float", v); }
    bool getParameter(const std::string& k, std::vector<std::string>& v) const { return fetch(k, "get_parameter_vector_string", v); }
    bool hasGroup(const std::string& name) const {
        try { return !group(name).is_none(); } catch (const py::error_already_set&) { return false; } catch (const std::exception&) { return false; }
    }
    py::object group(const std::string& name) const {
        for (const char* g : {"get_group", "getGroup"})
            if (py::hasattr(m, g)) return m.attr(g)(name);
        return m[py::str(name)];
    }

private:
    template <class T>
    bool fetch(const std::string& k, const char* blfGetter, T& v) const {
        try {
            py::object o;
            if (py::hasattr(m, blfGetter)) o = m.attr(blfGetter)(k);
            else if (py::hasattr(m, "getParameter")) {
                o = m.attr("getParameter")(k);
                if (py::isinstance<py::tuple>(o)) {
                    py::tuple t = o;
                    if (!t[0].cast<bool>()) return false;
                    o = t[1];
                }
            } else {
                if (!m.contains(py::str(k))) return false;
                o = m[py::str(k)];
            }
            if (o.is_none()) return false;
            v = o.cast<T>();
            return true;
        } catch (const py::error_already_set&) {
            return false;
        } catch (const py::cast_error&) {
            return false;
        }
    }
    py::object m;
};

py::object& default_loader() { static py::object* l = new py::object(py::none()); return *l; }

// a trajectory group as a mapping {track: array, "fps": n}: the group itself when it already is one, or what the loader
// returns for the group's `trajectoryFile`
py::object resolve_group(const py::object& group, const py::object& loader) {
    std::string file;
    if (PyParams(group).getParameter("trajectoryFile", file)) {
        const py::object& l = loader.is_none() ? default_loader() : loader;
        if (l.is_none())
            throw std::invalid_argument("group holds trajectoryFile '" + file + "' but no trajectory loader is set "
                                        "(setTrajectoryLoader / set_trajectory_loader)");
        return l(file);
    }
    return group;
}

int fps_of(const py::object& group, int fallback) {
    try {
        if (py::hasattr(group, "get_parameter_int")) return group.attr("get_parameter_int")("fps").cast<int>();
        if (group.contains(py::str("fps"))) return group[py::str("fps")].cast<int>();
    } catch (const py::error_already_set&) {}
    return fallback;
}
py::object item(const py::object& group, const char* k) {
    return group.contains(py::str(k)) ? py::object(group[py::str(k)]) : py::none();
}
void add_track(Trajectory& t, const char* name, const py::object& o, int dim, int n_default, int fps, int des) {
    if (o.is_none()) {
        std::vector<double> z(size_t(n_default) * dim, 0.0);
        t.add(name, z.data(), n_default, dim, fps, des);
        return;
    }
    Arr a = Arr::ensure(o);
    if (!a) throw std::invalid_argument(std::string("trajectory '") + name + "' must be an array");
    if (a.size() % dim != 0) throw std::invalid_argument(std::string("trajectory '") + name + "' has the wrong shape");
    t.add(name, a.data(), int(a.size() / dim), dim, fps, des);
}

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

// both front ends behind one Python class
class PyVariableSamplingMPC {
public:
    using Surface = vsmpc_host::VariableSamplingMPCT<PyQPInput>;

    // MPCPyBindings.cpp:24-32
    bool configureReference(py::object parametersHandler, py::object mpcInput, int device) {
        PyParams top(parametersHandler);
        PyParams params(top.hasGroup("VS_MPC_CONFIG") ? top.group("VS_MPC_CONFIG") : parametersHandler);
        double periodMPC = 0.0, periodLarge = 0.0;
        if (!params.getParameter("periodMPC", periodMPC) || !params.getParameter("periodMPCLargeSteps", periodLarge)) {
            py::print("[VariableSamplingMPC::configure] Parameter 'periodMPC' / 'periodMPCLargeSteps' not found in the config file.");
            return false;
        }
        auto position = std::make_shared<Trajectory>();
        auto alpha = std::make_shared<Trajectory>();
        try {
            const py::object pt = resolve_group(params.group("POSITION_TRAJECTORY"), m_loader);
            const py::object tm = resolve_group(params.group("TRAJECTORY_MANAGER"), m_loader);
            const int desPos = int(1.0 / periodLarge), desAlpha = int(1.0 / periodMPC);   // costsVSMPC.cpp:68, systemDynamicsVSMPC.cpp:272
            const int fpsPos = fps_of(pt, 10), fpsAlpha = fps_of(tm, 10);
            Arr pos = Arr::ensure(item(pt, "positionCoM"));
            if (!pos) throw std::invalid_argument("POSITION_TRAJECTORY/positionCoM missing");
            const int n = int(pos.size() / 3);
            add_track(*position, "positionCoM", pos, 3, n, fpsPos, desPos);
            add_track(*position, "velocityCoM", item(pt, "velocityCoM"), 3, n, fpsPos, desPos);
            add_track(*position, "RPY", item(pt, "RPY"), 3, n, fpsPos, desPos);
            add_track(*position, "RPYDot", item(pt, "RPYDot"), 3, n, fpsPos, desPos);
            add_track(*alpha, "alphaGravity", item(tm, "alphaGravity"), 1, 1, fpsAlpha, desAlpha);
        } catch (const std::exception& e) {
            py::print("[VariableSamplingMPC::configure]", e.what());
            return false;
        }
        m_ref = std::make_unique<Surface>();
        m_ref->setTrajectories(position, alpha);
        m_ref->setDevice(device);
        m_ref->setFusedTick(m_fused);
        PyQPInput qp(mpcInput);
        bool ok = false;
        try {
            ok = m_ref->configure(params, qp);
        } catch (const std::exception& e) {
            py::print("[VariableSamplingMPC::configure]", e.what());
            return false;
        }
        if (!ok) {
            py::print("[VariableSamplingMPC::configure]", m_ref->getLastMessage());
            m_ref.reset();
        }
        m_rec.reset();
        return ok;
    }
    // record level
    bool configureRecord(const py::dict& params, py::object jointPos, py::object rpy0, int device) {
        const vsmpc_config c = config_from_dict(params);
        std::vector<double> q(vsmpc_host::kRobotJoints, 0.0), r(3, 0.0);
        if (!jointPos.is_none()) q = jointPos.cast<std::vector<double>>();
        if (!rpy0.is_none()) r = rpy0.cast<std::vector<double>>();
        if (q.size() != size_t(vsmpc_host::kRobotJoints) || r.size() != 3) return false;
        m_rec = std::make_unique<vsmpc_host::VariableSamplingMPC>();
        m_ref.reset();
        if (!m_rec->configure(c, q.data(), r.data(), device)) { m_rec.reset(); return false; }
        return true;
    }
    bool update(py::object mpcInput, bool applyTickState) {
        if (m_ref) {                                           // MPCPyBindings.cpp:37
            PyQPInput qp(mpcInput);
            return m_ref->update(qp);
        }
        if (!m_rec) return false;
        Arr rec = Arr::ensure(mpcInput);
        if (!rec || rec.size() != m_rec->inputDoubles()) return false;
        return m_rec->update(rec.data(), applyTickState);
    }
    bool solveMPC() { return m_ref ? m_ref->solveMPC() : (m_rec ? m_rec->solveMPC() : true); }
    int status() const { return m_ref ? m_ref->getQPProblemStatus() : (m_rec ? m_rec->getQPProblemStatus() : 0); }
    const vsmpc_host::TickMachine& tick() const {
        if (m_ref) return m_ref->tickMachine();
        if (!m_rec) throw std::runtime_error("VariableSamplingMPC is not configured");
        return m_rec->tickState();
    }
    py::array_t<double> solutionInputs() const {
        const auto& x = tick().solution();
        const size_t off = size_t(VSMPC_N_STATES) * (tick().parameters().cfg.n_iter + 1);
        return py::array_t<double>(x.size() - off, x.data() + off);
    }
    py::array_t<double> final3(int off) const { return py::array_t<double>(3, tick().finalState().data() + off); }
    void setTrajectoryLoader(py::object loader) { m_loader = std::move(loader); }
    void setFusedTick(bool fused) { m_fused = fused; }

private:
    py::object m_loader = py::none();
    bool m_fused = true;
    std::unique_ptr<Surface> m_ref;
    std::unique_ptr<vsmpc_host::VariableSamplingMPC> m_rec;
};

}  // namespace

PYBIND11_MODULE(bindingsMPC, m) {
    m.doc() = "MI355X-backed drop-in for momentum_based_mpc.bindingsMPC (VariableSamplingMPC only)";
    m.def("set_trajectory_loader", [](py::object loader) { default_loader() = std::move(loader); }, py::arg("loader"),
          "loader(trajectoryFile) -> {variable: array, 'fps': int}; used when a handler group holds `trajectoryFile`");
    py::class_<PyVariableSamplingMPC>(m, "VariableSamplingMPC")
        .def(py::init<>())
        .def("configure", &PyVariableSamplingMPC::configureReference, py::arg("parametersHandler"), py::arg("mpcInput"),
             py::arg("device") = 0)                                              // MPCPyBindings.cpp:24-32 (tried first)
        .def("configureRecord", &PyVariableSamplingMPC::configureRecord, py::arg("parametersHandler"), py::arg("jointPositions"),
             py::arg("initialRPY"), py::arg("device") = 0)                       // record level: dict, 23 joints, RPY
        .def("setTrajectoryLoader", &PyVariableSamplingMPC::setTrajectoryLoader, py::arg("loader"))
        .def("setFusedTick", &PyVariableSamplingMPC::setFusedTick, py::arg("fused"))
        .def("update", &PyVariableSamplingMPC::update, py::arg("mpcInput"), py::arg("applyTickState") = false)
        .def("solveMPC", &PyVariableSamplingMPC::solveMPC)
        .def("getQPProblemStatus", &PyVariableSamplingMPC::status)
        .def("getMPCSolution", &PyVariableSamplingMPC::solutionInputs)
        .def("getJointsReferencePosition", [](PyVariableSamplingMPC& s) { return vec(s.tick().jointsPositionReference()); })   // :40-46
        .def("getThrottleReference", [](PyVariableSamplingMPC& s) { return vec(s.tick().throttle()); })
        .def("getThrustReference", [](PyVariableSamplingMPC& s) { return vec(s.tick().thrust()); })
        .def("getThrustDotReference", [](PyVariableSamplingMPC& s) { return vec(s.tick().thrustDot()); })
        .def("getFinalCoMPosition", [](PyVariableSamplingMPC& s) { return s.final3(0); })
        .def("getFinalLinMom", [](PyVariableSamplingMPC& s) { return s.final3(3); })
        .def("getFinalRPY", [](PyVariableSamplingMPC& s) { return s.final3(6); })
        .def("getFinalAngMom", [](PyVariableSamplingMPC& s) { return s.final3(9); })
        .def("getRecord", [](PyVariableSamplingMPC& s) { return vec(s.tick().record()); })   // the vsmpc_input record of the last update()
        .def("getNStatesMPC", [](PyVariableSamplingMPC&) { return double(VSMPC_N_STATES); })
        .def("getNInputMPC", [](PyVariableSamplingMPC&) { return double(VSMPC_N_JOINTS + VSMPC_N_THRUSTS); });
}
