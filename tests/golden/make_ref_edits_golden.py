"""Edits files written by the REFERENCE's own writer: Testbed::save_edits (src/testbed.cu:3190-3204) -> CageDeformation::to_json / AffineDuplication::to_json
-> GrowingSelection::to_json (growing_selection.cu:2459-2480) -> to_json of Cage (cage.h:100-119), TetMesh (tet_mesh.h:136-156), Mesh, AffineBoundingBox
(affine_bounding_box.cuh:134-143), BoundingBox and the Eigen / std::vector bindings (json_binding.h:27-321), compiled from /root/reference into
oracle/_ref/libref_json.so by oracle/Makefile (only possible where /root/reference is mounted; oracle/ref_json.cpp says what is a stand-in).  Run from the repo root:

    python tests/golden/make_ref_edits_golden.py

Writes tests/golden/ref_edits_golden.json.gz (the file, gzip'd) and ref_edits_golden.npz (the arrays that went INTO the reference's writer).  The pin
(tests/test_ref_pin.py::test_edits_reader_golden): nrs_edits_open on the file returns those arrays bit for bit.  Operators, in file order:
  0  cage_deformation of the test edit (lattice 6: 343 tet vertices, 1296 tets, 26 cage vertices) with MVC weights, gamma coordinates and the cage's membrane terms
  1  affine_duplication (rotated selection box, non-trivial translation / scale / rotation, hide_original on, correct_dir off)
  2  cage_deformation whose interpolation mesh carries no MVC weights (the reference writes `null` for an empty vector) -- values chosen to stress the number
     parser: denormals, -0, 2^-126, FLT_MAX, 16777217-style neighbours
  3  cage_deformation without an interpolation mesh (cage not yet tetrahedralised, growing_selection.cu:2477: the key is absent)"""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def inputs():
    from nerfshop_amd import synth, _abi
    e = synth.make_cage_edit(lattice_n=6)
    rng = np.random.default_rng(77)
    V, NC = e.vertices.shape[0], e.cage_vertices.shape[0]
    a = {"op0/cage_vertices": e.cage_deformed, "op0/cage_original_vertices": e.cage_vertices, "op0/cage_triangles": e.cage_triangles,
         "op0/vertices": e.vertices, "op0/original_vertices": e.original_vertices, "op0/tets": e.tets, "op0/mvc": e.mvc_weights,
         "op0/gamma": rng.uniform(0, 1, size=(V, NC)).astype(np.float32),
         "op0/inside_density": rng.uniform(0, 50, NC).astype(np.float32), "op0/outside_density": rng.uniform(0, 50, NC).astype(np.float32),
         "op0/inside_shs": rng.normal(0, 0.3, (NC, 27)).astype(np.float32), "op0/outside_shs": rng.normal(0, 0.3, (NC, 27)).astype(np.float32)}
    th = 0.7
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
    th2 = -0.3
    rot2 = np.array([[1, 0, 0], [0, np.cos(th2), -np.sin(th2)], [0, np.sin(th2), np.cos(th2)]], np.float32)
    a.update({"op1/selection_center": np.array([0.4, 0.55, 0.5], np.float32), "op1/selection_scale": np.array([0.1, 0.2, 0.15], np.float32),
              "op1/selection_rot": rot.T.reshape(-1).copy(),  # column-major
              "op1/selection_min": np.array([0.3, 0.35, 0.35], np.float32), "op1/selection_max": np.array([0.5, 0.75, 0.65], np.float32),
              "op1/translation": np.array([0.25, -0.125, 1e-3], np.float32), "op1/scale": np.array([1.5, 0.75, 1.0], np.float32),
              "op1/rotation": rot2.T.reshape(-1).copy(), "op1/hide_original": np.array(1, np.int32), "op1/correct_dir": np.array(0, np.int32)})
    odd = np.array([[0.0, -0.0, 1e-45], [1.17549435e-38, 3.4028235e38, -3.4028235e38], [16777217.0, 0.1, 1.0 / 3.0], [5e-324, 1e-40, 0.30000001192092896]], np.float32)
    a.update({"op2/cage_vertices": odd, "op2/cage_original_vertices": odd[::-1].copy(), "op2/cage_triangles": np.array([[0, 1, 2], [1, 2, 3]], np.uint32),
              "op2/vertices": odd * np.float32(0.5), "op2/original_vertices": odd, "op2/tets": np.array([[0, 1, 2, 3]], np.uint32)})
    a.update({"op3/cage_vertices": e.cage_deformed[:8].copy(), "op3/cage_original_vertices": e.cage_vertices[:8].copy(),
              "op3/cage_triangles": np.array([[0, 1, 2], [2, 3, 4], [5, 6, 7]], np.uint32)})
    return {k: np.ascontiguousarray(v) for k, v in a.items()}


def redump_with_nlohmann(raw):
    """bytes of a JSON file -> the bytes nlohmann/json 3.1.1 writes for the same value tree (`f << j << std::endl`), through oracle/_ref/json_redump"""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "a.json"), os.path.join(d, "b.json")
        open(a, "wb").write(raw)
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "json_redump"), a, b])
        return open(b, "rb").read()


def write_with_reference(path, a):
    from oracle import ref_json
    from nerfshop_amd import _abi
    op0 = ref_json.cage_op(a["op0/cage_vertices"], a["op0/cage_original_vertices"], a["op0/cage_triangles"], a["op0/vertices"], a["op0/original_vertices"], a["op0/tets"],
                           mvc=a["op0/mvc"], gamma=a["op0/gamma"], inside_density=a["op0/inside_density"], outside_density=a["op0/outside_density"],
                           inside_shs=a["op0/inside_shs"], outside_shs=a["op0/outside_shs"])
    ad = _abi.AffineDuplicationOp()
    for name in ("selection_center", "selection_scale", "selection_rot", "translation", "scale", "rotation"):
        setattr(ad, name, (type(getattr(ad, name)))(*[float(v) for v in a["op1/" + name]]))
    ad.hide_original, ad.correct_dir = int(a["op1/hide_original"].reshape(-1)[0]), int(a["op1/correct_dir"].reshape(-1)[0])
    op1 = ref_json.affine_op(ad, a["op1/selection_min"], a["op1/selection_max"])
    op2 = ref_json.cage_op(a["op2/cage_vertices"], a["op2/cage_original_vertices"], a["op2/cage_triangles"], a["op2/vertices"], a["op2/original_vertices"], a["op2/tets"])
    op3 = ref_json.cage_op(a["op3/cage_vertices"], a["op3/cage_original_vertices"], a["op3/cage_triangles"])
    ref_json.save_edits(path, [op0, op1, op2, op3])


if __name__ == "__main__":
    a = inputs()
    tmp = os.path.join(ROOT, "tests", "golden", "_ref_edits.json")
    write_with_reference(tmp, a)
    raw = open(tmp, "rb").read()
    os.remove(tmp)
    out = os.path.join(ROOT, "tests", "golden", "ref_edits_golden.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(raw)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_edits_golden.npz"), **a)
    print("wrote", out, len(raw), "->", os.path.getsize(out), "bytes")
    # the same value tree as the REAL nlohmann/json prints it (oracle/ref_json_redump.cpp: this image's 3.1.1 reads the file above and writes it back the way
    # Testbed::save_edits does -- shortest round-trip floats instead of the stand-in's 17 digits)
    redump = os.path.join(ROOT, "oracle", "_ref", "json_redump")
    if os.path.exists(redump):
        raw2 = redump_with_nlohmann(raw)
        out2 = os.path.join(ROOT, "tests", "golden", "ref_edits_golden_nlohmann.json.gz")
        with gzip.GzipFile(out2, "wb", mtime=0) as f:
            f.write(raw2)
        print("wrote", out2, len(raw2), "->", os.path.getsize(out2), "bytes")
