"""Writes tests/golden/ref_pin_golden.npz: outputs of the REFERENCE's own render-path code (oracle/_ref/libref_render.so: the reference's
headers, src/common_nerf.cu and the kernels cut out of testbed_nerf.cu / cage_deformation.cu / tet_mesh.cu / affine_duplication.cu, compiled
as host code -- see oracle/ref_render.cpp) on the seeded cases of tests/ref_pin_cases.py.  Small results are stored whole, 10^5-input
results as SHA-256 of their bytes plus the first 64 rows.  Only possible where /root/reference is mounted:

    make -C oracle && python tests/golden/make_ref_pin_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_pin_cases as cases  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    assert ref.available(), "oracle/_ref/libref_render.so is missing: run `make -C oracle` where /root/reference is mounted"
    out = {}

    def store_hashed(prefix, d):
        for name, arrays in d.items():
            out[f"{prefix}/{name}/sha"] = cases.sha(*arrays)
            for k, a in enumerate(arrays):
                out[f"{prefix}/{name}/head{k}"] = np.ascontiguousarray(a)[:64].copy()

    scenes = cases.Scenes()
    store_hashed("probe", cases.run_probes("ref"))
    store_hashed("probe", {"pixel_to_ray": cases.pixel_to_ray_case(scenes.get("lego"), "ref")})
    store_hashed("op", cases.operator_cases(scenes, "ref"))
    store_hashed("authoring", cases.authoring_cases(scenes, "ref"))
    store_hashed("bitfield", {"grids": cases.bitfield_case(scenes, "ref")})
    store_hashed("refresh", cases.refresh_cases(scenes, "ref"))
    store_hashed("selection", cases.selection_cases(scenes, "ref"))
    store_hashed("grid_eval", cases.grid_eval_cases(scenes, "ref"))
    store_hashed("poisson_boundary", cases.poisson_boundary_cases(scenes, "ref"))
    store_hashed("accumulate", cases.accumulate_cases("ref"))
    for case in cases.FRAME_CASES:
        f, d, s, st = cases.render_case(scenes, case, "ref")
        out[f"frame/{case[0]}/frame"], out[f"frame/{case[0]}/depth"], out[f"frame/{case[0]}/steps"], out[f"frame/{case[0]}/stats"] = f, d, s.astype(np.uint16), st
        print(case[0], "hit", st[0], "composited", st[1])
    for k, case in enumerate(cases.STREAM_CASES):
        coords, t_after, cnt, odt = cases.stream_case(scenes, case, "ref")
        out[f"stream/{k}/sha"] = cases.sha(coords, t_after, cnt, odt)
        out[f"stream/{k}/count"] = cnt.astype(np.uint16)
        print("stream", k, case, "samples", int(cnt.sum()), "rays", int((cnt > 0).sum()))
    path = os.path.join(ROOT, "tests", "golden", "ref_pin_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
