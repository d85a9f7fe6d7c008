// Text point-cloud input on either side of the path: one point per line, `x y z r g b`, colours
// discarded — the format of the reference's fixture and the behaviour of its test loader
// (/root/reference/tst/point2point.cpp:125-138, tst/data/fachada.txt).  Returns packed xyz, the
// layout the device models take.
#pragma once

#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace moptimizer {
namespace io {

template <typename Scalar = double>
inline std::vector<Scalar> loadXyzRgbText(const std::string &path) {
  std::ifstream in(path);
  if (!in.is_open()) throw std::runtime_error("not a file! exiting");
  std::vector<Scalar> xyz;
  double x, y, z, discard;
  // stops at the first line that does not parse, like the reference's `while (file >> ...)`
  while (in >> x >> y >> z >> discard >> discard >> discard) {
    xyz.push_back(static_cast<Scalar>(x));
    xyz.push_back(static_cast<Scalar>(y));
    xyz.push_back(static_cast<Scalar>(z));
  }
  return xyz;
}

}  // namespace io
}  // namespace moptimizer
