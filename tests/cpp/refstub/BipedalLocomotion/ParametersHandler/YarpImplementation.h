// tests/cpp/refstub/.../YarpImplementation.h -- TEST INFRASTRUCTURE, not BLF (see ../../README.md).
// A map-backed IParametersHandler under the name of BLF's YARP-backed one (the type TrajectoryManager::configure takes,
// utils/include/TrajectoryManager.h:73-75); values are put in by the tests through test*().
#ifndef REFSTUB_BLF_YARP_IMPLEMENTATION_H
#define REFSTUB_BLF_YARP_IMPLEMENTATION_H
#include <map>

#include <BipedalLocomotion/ParametersHandler/IParametersHandler.h>

namespace BipedalLocomotion {
namespace ParametersHandler {

class YarpImplementation : public IParametersHandler {
public:
    bool getParameter(const std::string& k, int& v) const override { return scalar(k, v); }
    bool getParameter(const std::string& k, double& v) const override { return scalar(k, v); }
    bool getParameter(const std::string& k, bool& v) const override { return scalar(k, v); }
    bool getParameter(const std::string& k, std::string& v) const override {
        auto it = m_strings.find(k);
        if (it == m_strings.end()) return false;
        v = it->second;
        return true;
    }
    bool getParameter(const std::string&, std::vector<bool>&) const override { return false; }
    bool getParameter(const std::string&, GenericContainer::Vector<int>::Ref) const override { return false; }
    bool getParameter(const std::string& k, GenericContainer::Vector<double>::Ref v) const override { return list(m_vectors, k, v); }
    bool getParameter(const std::string& k, GenericContainer::Vector<std::string>::Ref v) const override { return list(m_stringLists, k, v); }
    bool setGroup(const std::string& name, shared_ptr newGroup) override { m_groups[name] = newGroup; return true; }
    weak_ptr getGroup(const std::string& name) const override {
        auto it = m_groups.find(name);
        return it == m_groups.end() ? weak_ptr() : weak_ptr(it->second);
    }

    // test loaders
    void testSet(const std::string& k, double v) { m_scalars[k] = v; }
    void testSet(const std::string& k, const std::string& v) { m_strings[k] = v; }
    void testSet(const std::string& k, const char* v) { m_strings[k] = v; }
    void testSet(const std::string& k, const std::vector<double>& v) { m_vectors[k] = v; }
    void testSet(const std::string& k, const std::vector<std::string>& v) { m_stringLists[k] = v; }
    void testErase(const std::string& k) { m_scalars.erase(k); m_strings.erase(k); m_vectors.erase(k); m_stringLists.erase(k); }

private:
    template <class T>
    bool scalar(const std::string& k, T& v) const {
        auto it = m_scalars.find(k);
        if (it == m_scalars.end()) return false;
        v = T(it->second);
        return true;
    }
    template <class M, class V>
    static bool list(const M& m, const std::string& k, const V& v) {
        auto it = m.find(k);
        if (it == m.end()) return false;
        v.resize(it->second.size());
        for (std::size_t i = 0; i < it->second.size(); ++i) v[i] = it->second[i];
        return true;
    }
    std::map<std::string, double> m_scalars;
    std::map<std::string, std::string> m_strings;
    std::map<std::string, std::vector<double>> m_vectors;
    std::map<std::string, std::vector<std::string>> m_stringLists;
    std::map<std::string, shared_ptr> m_groups;
};

}  // namespace ParametersHandler
}  // namespace BipedalLocomotion
#endif
