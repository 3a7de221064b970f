// oracle/shim/boost/unordered_map.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.
#pragma once
#include <functional>
#include <unordered_map>
namespace boost {
template <typename K, typename V, typename H = std::hash<K>, typename E = std::equal_to<K> > using unordered_map = std::unordered_map<K, V, H, E>;
template <typename T> inline void hash_combine(std::size_t& seed, const T& v) { seed ^= std::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2); }
template <typename T> struct hash : std::hash<T> {};
}  // namespace boost
