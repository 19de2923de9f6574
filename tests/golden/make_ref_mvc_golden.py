"""Golden vectors from the REFERENCE's own code: MVC3D::computeCoordinatesCustomCode of
/root/reference/include/neural-graphics-primitives/editing/tools/mvc.h, compiled as oracle/_ref/libref_mvc.so by
oracle/Makefile (only possible where /root/reference is mounted).  Run from the repo root:

    python tests/golden/make_ref_mvc_golden.py

Writes tests/golden/ref_mvc_golden.npz: the cage, the query points and the reference's weights / labels.  The points are
the tet-lattice vertices of the test edit plus the special cases of the routine: a cage vertex (early exit), points on
cage faces (the 2-D barycentric branch), points outside the cage, random interior points."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ref_mvc(cage_v, cage_t, points):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_mvc.so"))
    lib.ref_mvc_compute.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.ref_mvc_compute.restype = None
    cv = np.ascontiguousarray(cage_v, np.float32)
    ct = np.ascontiguousarray(cage_t, np.uint32)
    pts = np.ascontiguousarray(points, np.float32)
    w = np.zeros((pts.shape[0], cv.shape[0]), np.float32)
    labels = np.zeros(pts.shape[0], np.uint8)
    lib.ref_mvc_compute(cv.ctypes.data, cv.shape[0], ct.ctypes.data, ct.shape[0], pts.ctypes.data, pts.shape[0], w.ctypes.data, labels.ctypes.data)
    return w, labels


def inputs():
    from nerfshop_amd import synth
    e = synth.make_cage_edit(lattice_n=6)
    cv, ct = e.cage_vertices, e.cage_triangles
    rng = np.random.default_rng(2024)
    lo, hi = cv.min(0), cv.max(0)
    special = [cv[3], cv[10]]                                                  # exactly on cage vertices
    tri = cv[ct[5]]
    special.append(tri.mean(0))                                                # centroid of a cage triangle
    special.append((0.2 * tri[0] + 0.5 * tri[1] + 0.3 * tri[2]))               # another point on that face
    special += [lo - 0.05, hi + np.float32(0.1), 0.5 * (lo + hi) + np.array([0, (hi - lo)[1], 0], np.float32)]  # outside
    pts = np.concatenate([e.original_vertices, np.array(special, np.float32), rng.uniform(lo, hi, size=(200, 3)).astype(np.float32)])
    return np.ascontiguousarray(cv, np.float32), np.ascontiguousarray(ct, np.uint32), np.ascontiguousarray(pts, np.float32)


if __name__ == "__main__":
    cv, ct, pts = inputs()
    w, labels = ref_mvc(cv, ct, pts)
    out = os.path.join(ROOT, "tests", "golden", "ref_mvc_golden.npz")
    np.savez_compressed(out, cage_vertices=cv, cage_triangles=ct, points=pts, weights=w, labels=labels)
    print("wrote", out, "points", pts.shape[0], "labels==0:", int((labels == 0).sum()), "size", os.path.getsize(out))
