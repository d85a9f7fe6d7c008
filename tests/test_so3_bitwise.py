"""The transforms the product derives from x on the host are BIT-identical to the oracle's.

Forward differences (linearization.h:78-105) divide R(x + h e_j) - R(x) by h_j = sqrt(eps) |x_j|
~ 1e-8 |x_j|, so a 1-ulp difference between the two statements of Rodrigues' formula would move a
whole Jacobian column by eps / h_j, coherently over all points — the residual numeric-mode
disagreement of round 1 (a compiler fusing multiply-adds in one of them).  Both are now rounded
operation by operation, so numeric-mode parity holds at the 1e-6 bar for every x, not only where
|x_j| is large.  No GPU: mopt_se3_from_params is host code of the shipped library.
"""
import subprocess

import numpy as np

from tests import datasets as ds
from tests import oracle_binding as ob


def _grid():
    rng = np.random.default_rng(7)
    xs = [ds.X_ZERO, ds.X_GENERIC]
    for scale in (1e-9, 1e-6, 1e-4, 1e-2, 0.1, 0.7, 3.0):  # |x_j| decades
        xs += list(rng.normal(0.0, scale, size=(300, 6)))
    # pure rotations about one axis, near pi, tiny angles around the 10 eps switch of so3.cpp:47
    for a in (1e-16, 2.2e-15, 2.3e-15, 1e-8, np.pi - 1e-9, np.pi, 2 * np.pi, 6.5):
        for k in range(3):
            x = np.zeros(6)
            x[3 + k] = a
            xs.append(x)
    return xs


def test_library_and_oracle_transforms_agree_bitwise(oracle):
    import moptimizer_0_amd as mo
    checked = 0
    for x in _grid():
        T, Tp, h = mo.capi.se3_from_params(x, with_steps=True)
        To, Tpo, ho = oracle.se3_from_x(x, with_steps=True)
        assert T.tobytes() == To.tobytes(), x
        assert h.tobytes() == ho.tobytes(), x
        for j in range(6):
            assert Tp[j].tobytes() == Tpo[j].tobytes(), (x, j)
        checked += 7
    assert checked > 14000


def test_headers_agree_bitwise_under_aggressive_flags(tmp_path):
    """The same two headers compiled into ONE program with multiply-add fusion enabled
    (-O3 -march=x86-64-v3, GCC's default -ffp-contract=fast): still bit-identical."""
    src = tmp_path / "so3_bits.cpp"
    src.write_text(r'''
#include "moptimizer_amd/so3.hpp"
#include "so3_ref.hpp"
#include <cstdio>
#include <cstring>
#include <random>
int main() {
  std::mt19937_64 g(1);
  std::normal_distribution<double> n(0, 0.7);
  long bad = 0;
  for (int it = 0; it < 20000; ++it) {
    double x[6];
    for (auto &v : x) v = n(g);
    if (it % 3 == 0) for (auto &v : x) v *= 1e-3;
    for (int j = -1; j < 6; ++j) {
      double xp[6];
      std::memcpy(xp, x, sizeof x);
      if (j >= 0) xp[j] += 1.4901161193847656e-8 * std::fabs(x[j]);
      const auto T = moptimizer::so3::rigidFrom6DOF<double>(xp);
      double M[16];
      oracle::so3::convert6DOFParameterToMatrix<double>(xp, M);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) bad += std::memcmp(&T.m[r * 4 + c], &M[c * 4 + r], 8) != 0;
    }
  }
  std::printf("%ld\n", bad);
  return bad != 0;
}
''')
    exe = tmp_path / "so3_bits"
    subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17",
                           "-I", ds.ROOT + "/include", "-I", ds.ROOT + "/oracle",
                           str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)]).decode().strip() == "0"
