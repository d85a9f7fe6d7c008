"""Pack the reference's point-cloud fixture into a small binary.

Input : /root/reference/tst/data/fachada.txt  (29 310 lines `x y z r g b`, the data file the
        reference's tst/point2point.cpp:88,125-138 loads; colours are discarded there too).
Output: tests/golden/fachada_xyz_1e8.npz  -- int32 array [29310, 3] = round(coordinate * 1e8).

The text carries 8 decimals and |coordinate| < 21.47, so the scaled integers fit int32, and
`q / 1e8` (one correctly-rounded division) reproduces the double that the reference's
`cloud_file >> x` parse yields, bit for bit (checked below).  Run here only: the reference tree
does not exist on the GPU box.
"""
import os
import numpy as np

SRC = "/root/reference/tst/data/fachada.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fachada_xyz_1e8.npz")

if __name__ == "__main__":
    xyz = np.loadtxt(SRC)[:, :3]
    q = np.round(xyz * 1e8).astype(np.int64)
    assert np.abs(q).max() < 2**31
    assert np.array_equal(q / 1e8, xyz), "fixed-point round trip must be exact"
    np.savez_compressed(DST, xyz_1e8=q.astype(np.int32))
    print(DST, q.shape, os.path.getsize(DST), "bytes")
