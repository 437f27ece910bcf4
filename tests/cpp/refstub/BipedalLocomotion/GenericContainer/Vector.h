// tests/cpp/refstub/.../GenericContainer/Vector.h -- TEST INFRASTRUCTURE, not BLF (see ../../README.md).
// BLF's IParametersHandler takes vector parameters as GenericContainer::Vector<T>::Ref, a resizable view that any
// container of T with data() / size() / resize() converts to implicitly (std::vector<double>, Eigen::VectorXd, ...).
// Recalled from BLF's public interface; its source is not in this image.
#ifndef REFSTUB_BLF_GENERIC_CONTAINER_VECTOR_H
#define REFSTUB_BLF_GENERIC_CONTAINER_VECTOR_H
#include <cstddef>
#include <functional>
#include <type_traits>
#include <utility>

namespace BipedalLocomotion {
namespace GenericContainer {

template <class T>
class Vector {
public:
    using Ref = const Vector<T>;   // by value: a temporary view converted from the caller's container

    template <class Container,
              std::enable_if_t<std::is_same<std::remove_cv_t<std::remove_reference_t<decltype(*std::declval<Container&>().data())>>, T>::value
                                   && !std::is_const<Container>::value, int> = 0>
    Vector(Container& c)
        : m_resize([&c](std::size_t n) { c.resize(n); return c.data(); }), m_data(c.data()), m_size(std::size_t(c.size())) {}

    void resize(std::size_t n) const { m_data = m_resize(n); m_size = n; }
    T& operator[](std::size_t i) const { return m_data[i]; }
    std::size_t size() const { return m_size; }

private:
    std::function<T*(std::size_t)> m_resize;
    mutable T* m_data;
    mutable std::size_t m_size;
};

}  // namespace GenericContainer
}  // namespace BipedalLocomotion
#endif
