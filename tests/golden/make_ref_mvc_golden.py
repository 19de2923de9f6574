"""Golden vectors from the REFERENCE's own code: MVC3D::computeCoordinatesCustomCode of
/root/reference/include/neural-graphics-primitives/editing/tools/mvc.h instantiated with point_t = Eigen::Vector3f (growing_selection.h:90),
compiled into oracle/_ref/libref_render.so by oracle/Makefile (only possible where /root/reference is mounted).  Run from the repo root:

    python tests/golden/make_ref_mvc_golden.py

Writes tests/golden/ref_mvc_golden.npz: the cage, the query points and the reference's weights / labels.  The points are
the tet-lattice vertices of the test edit plus the special cases of the routine: a cage vertex (early exit), points on
cage faces (the 2-D barycentric branch), points outside the cage, random interior points."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ref_mvc(cage_v, cage_t, points):
    from oracle import ref
    return ref.mvc_compute(cage_v, cage_t, points)


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
