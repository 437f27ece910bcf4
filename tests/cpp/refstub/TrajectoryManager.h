// tests/cpp/refstub/TrajectoryManager.h -- TEST INFRASTRUCTURE (see README.md): utils/include/TrajectoryManager.h:62-110.
// configure() reads `trajectoryFile` from group TRAJECTORY_MANAGER like the reference (TrajectoryManager.cpp:40-64) and
// then, instead of opening a MAT file (matio is absent), takes the arrays a test registered under that file name.  The
// up-sampling / cursor arithmetic is NOT restated here: it is delegated to vsmpc_host::Trajectory, whose agreement with
// TrajectoryManager.cpp:23-39,142-167 is what tests/tick_model.py pins.  What this class adds is the reference's
// INTERFACE: getCurrentValue returns an Eigen::VectorXd BY VALUE, throws on an unknown key (map::at), has no has().
#ifndef REFSTUB_TRAJECTORY_MANAGER_H
#define REFSTUB_TRAJECTORY_MANAGER_H
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <BipedalLocomotion/ParametersHandler/YarpImplementation.h>
#include <iDynTree/EigenHelpers.h>

#include "VariableSamplingMPC.hpp"   // vsmpc_host::Trajectory (the arithmetic)

class TrajectoryManager {
public:
    bool configure(std::shared_ptr<BipedalLocomotion::ParametersHandler::YarpImplementation> parametersHandler, int des_fps) {
        auto group = parametersHandler->getGroup("TRAJECTORY_MANAGER").lock();
        std::string trajectoryName;
        if (!group || !group->getParameter("trajectoryFile", trajectoryName)) return false;
        return loadTrajectoryFromFile(trajectoryName, des_fps);
    }
    bool loadTrajectoryFromFile(const std::string& trajectoryName, int des_fps) {
        auto it = testFiles().find(trajectoryName);
        if (it == testFiles().end()) return false;
        for (const auto& t : it->second.tracks) {
            m_impl.add(t.name, t.samples.data(), int(t.samples.size()) / t.dim, t.dim, it->second.fps, des_fps);
            m_dims[t.name] = t.dim;
        }
        return true;
    }
    bool advanceTrajectory() { return m_impl.advanceTrajectory(); }
    int getTrajectoryIndex() const { return m_impl.getTrajectoryIndex(); }
    Eigen::VectorXd getCurrentValue(std::string key) const {
        const int dim = m_dims.at(key);   // std::out_of_range on an unknown key, like trajectories_map.at(key)
        const double* p = m_impl.getCurrentValue(key);
        Eigen::VectorXd out(dim);
        for (int i = 0; i < dim; ++i) out(i) = p[i];
        return out;
    }

    // ---- test registry standing in for the MAT files
    struct TestTrack { std::string name; int dim; std::vector<double> samples; };
    struct TestFile { int fps = 10; std::vector<TestTrack> tracks; };
    static std::map<std::string, TestFile>& testFiles() { static std::map<std::string, TestFile> f; return f; }

private:
    vsmpc_host::Trajectory m_impl;
    std::map<std::string, int> m_dims;
};
#endif
