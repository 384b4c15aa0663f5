"""ORACLE — test infrastructure only (see each source file's header).

CPU restatements of the reference algorithms used as the parity checker by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never imported by the product
package (from-voxel-to-point_amd/)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


_f32, _i32, _i64 = ctypes.c_float, ctypes.c_int32, ctypes.c_int64


def points_to_voxel(points, voxel_size, coors_range, max_points, max_voxels, scratch_map=None):
    """voxel_generator.py:75-133 restated (reverse_index=True). Returns (voxels, coors, num_points)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rng = np.ascontiguousarray(coors_range, dtype=np.float32)
    n, ndim = points.shape
    voxels = np.zeros((max_voxels, max_points, ndim), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    f = lib().oracle_points_to_voxel
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(_f32), _i64, ctypes.c_int, ctypes.POINTER(_f32), ctypes.POINTER(_f32), ctypes.c_int,
                  ctypes.c_int, ctypes.POINTER(_f32), ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.POINTER(_i32)]
    sm = _p(scratch_map, _i32) if scratch_map is not None else None
    m = f(_p(points, _f32), n, ndim, _p(vs, _f32), _p(rng, _f32), max_points, max_voxels, _p(voxels, _f32),
          _p(coors, _i32), _p(num, _i32), sm)
    if m < 0:
        raise MemoryError("oracle_points_to_voxel")
    return voxels[:m], coors[:m], num[:m]
