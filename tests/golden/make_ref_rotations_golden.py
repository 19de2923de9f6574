"""Golden vectors from the REFERENCE's own per-tet rotations: the loop of TetMesh::update_local_rotations (tet_mesh.cu:49-70) and
include/neural-graphics-primitives/editing/tools/svd3.h (svd_eigen), compiled into oracle/_ref/libref_render.so (oracle/ref_render.cpp).
Run from the repo root where /root/reference is mounted:   python tests/golden/make_ref_rotations_golden.py
Writes tests/golden/ref_rotations_golden.npz: deformed / canonical vertices, tets, and the per-tet R = U V^T."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ref_rotations(vertices, original, tets):
    from oracle import ref
    return ref.local_rotations(vertices, original, tets)


if __name__ == "__main__":
    from nerfshop_amd import synth
    e = synth.make_cage_edit(lattice_n=6)
    # a second, harsher deformation of the same mesh (large twist + shear) so that the sort / sign-flip branches are exercised
    rng = np.random.default_rng(9)
    c = e.original_vertices.mean(0)
    th = 1.1
    Q = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    harsh = ((e.original_vertices - c) @ Q.T * np.array([1.4, 0.7, 1.0], np.float32) + c + rng.normal(0, 0.004, e.vertices.shape)).astype(np.float32)
    verts = np.concatenate([e.vertices, harsh])
    orig = np.concatenate([e.original_vertices, e.original_vertices])
    tets = np.concatenate([e.tets, e.tets + e.vertices.shape[0]]).astype(np.uint32)
    R = ref_rotations(verts, orig, tets)
    out = os.path.join(ROOT, "tests", "golden", "ref_rotations_golden.npz")
    np.savez_compressed(out, vertices=verts, original_vertices=orig, tets=tets, rotations=R)
    print("wrote", out, "tets", tets.shape[0], "size", os.path.getsize(out))
