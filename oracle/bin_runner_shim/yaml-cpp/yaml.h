// ORACLE test infrastructure (oracle/build_bin_runner.sh).  THIS IS NOT yaml-cpp: a stub of the four calls the reference's
// apps/cpp_runners/bin_runner.cpp makes on it (bin_runner.cpp:70-92) — YAML::LoadFile, node["key"], .as<double / int / bool>(),
// .as<std::vector<std::vector<double>>>() — for the flat `key : value  # comment` files of mad_icp/configurations/ (plus the
// one block list of flow lists, lidar_to_base).  yaml-cpp is not in this image and is not part of the hot path.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace YAML {

class Node {
  std::map<std::string, std::string> scalars_;
  std::map<std::string, std::vector<std::vector<double>>> lists_;
  std::string key_;
  const Node* root_ = nullptr;

  static std::string trim(const std::string& s) {
    const size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
    return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
  }
  const Node& root() const { return root_ ? *root_ : *this; }

 public:
  static Node load(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("yaml stub: cannot open " + path);
    Node n;
    std::string line, open_list;
    while (std::getline(in, line)) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line = line.substr(0, hash);
      const std::string t = trim(line);
      if (t.empty()) continue;
      if (t[0] == '-') {  // "- [a, b, c]": one row of the list opened by the last "key:" line
        if (open_list.empty()) throw std::runtime_error("yaml stub: list item without a key in " + path);
        std::string row = t.substr(1);
        for (char& c : row)
          if (c == '[' || c == ']' || c == ',') c = ' ';
        std::istringstream is(row);
        std::vector<double> v;
        double x;
        while (is >> x) v.push_back(x);
        n.lists_[open_list].push_back(v);
        continue;
      }
      const size_t colon = t.find(':');
      if (colon == std::string::npos) throw std::runtime_error("yaml stub: cannot parse '" + t + "' in " + path);
      const std::string key = trim(t.substr(0, colon)), value = trim(t.substr(colon + 1));
      if (value.empty()) {
        open_list = key;
        n.lists_[key];
      } else {
        open_list.clear();
        n.scalars_[key] = value;
      }
    }
    return n;
  }
  Node operator[](const std::string& key) const {
    Node c;
    c.key_ = key;
    c.root_ = &root();
    return c;
  }
  template <class T>
  T as() const;
};

template <>
inline double Node::as<double>() const {
  const auto it = root().scalars_.find(key_);
  if (it == root().scalars_.end()) throw std::runtime_error("yaml stub: no key " + key_);
  return std::strtod(it->second.c_str(), nullptr);
}
template <>
inline int Node::as<int>() const {
  return static_cast<int>(as<double>());
}
template <>
inline bool Node::as<bool>() const {
  const auto it = root().scalars_.find(key_);
  if (it == root().scalars_.end()) throw std::runtime_error("yaml stub: no key " + key_);
  const std::string& v = it->second;
  return v == "True" || v == "true" || v == "TRUE" || v == "yes" || v == "1";
}
template <>
inline std::vector<std::vector<double>> Node::as<std::vector<std::vector<double>>>() const {
  const auto it = root().lists_.find(key_);
  if (it == root().lists_.end()) throw std::runtime_error("yaml stub: no list " + key_);
  return it->second;
}

inline Node LoadFile(const std::string& path) { return Node::load(path); }

}  // namespace YAML
