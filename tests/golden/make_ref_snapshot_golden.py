"""A snapshot written by the REFERENCE's own writer: Testbed::save_snapshot (src/testbed.cu:3090-3113) over the reference's own configs/nerf/base.json, parsed where it
lies and run through its merge_parent_network_config (testbed.cu:86-97), with to_json(NerfDataset) (json_binding.h:136-160) -- compiled from /root/reference into
oracle/_ref/libref_json.so (oracle/ref_json.cpp says what is a stand-in: nlohmann/json itself, and tiny-cuda-nn's Trainer::serialize / GPUMemory -> json, whose three
keys `params_binary`, `params_type`, `n_params` therefore stay unpinned).  Run from the repo root:

    python tests/golden/make_ref_snapshot_golden.py

Writes tests/golden/ref_snapshot_golden.msgpack.gz (the file, gzip'd; log2_hashmap_size 12 and aabb_scale 4 so that it stays small and exercises the dataset's
aabb_scale) and ref_snapshot_golden.npz (the parameters and the density grid's non-zero entries that went in)."""
import ctypes as C
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CONFIG = "/root/reference/configs/nerf/base.json"
LOG2_T, AABB_SCALE = 12, 4


def inputs():
    from nerfshop_amd import synth, _abi
    desc = synth.model_desc(AABB_SCALE, log2_hashmap_size=LOG2_T)
    n = _abi.load().nrs_model_n_params(C.byref(desc))
    rng = np.random.default_rng(11)
    params = rng.integers(0, 0x7C00, size=n, dtype=np.uint16)            # every finite positive half pattern
    params[::7] |= 0x8000                                                # ... and negative ones
    idx = np.unique(rng.integers(0, 5 * 128 ** 3, 4000))
    val = rng.uniform(0.01, 40.0, idx.size).astype(np.float32)
    return desc, params, idx.astype(np.int64), val


def grid_from(idx, val):
    g = np.zeros(5 * 128 ** 3, np.float32)
    g[idx] = val
    return g


if __name__ == "__main__":
    from oracle import ref_json
    desc, params, idx, val = inputs()
    tmp = os.path.join(ROOT, "tests", "golden", "_ref_snapshot.msgpack")
    ref_json.save_snapshot(tmp, CONFIG, params, grid_from(idx, val), AABB_SCALE, log2_hashmap_size=LOG2_T, training_step=12345, loss=0.0123)
    raw = open(tmp, "rb").read()
    os.remove(tmp)
    out = os.path.join(ROOT, "tests", "golden", "ref_snapshot_golden.msgpack.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(raw)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_snapshot_golden.npz"), params=params, grid_idx=idx, grid_val=val,
                        per_level_scale=np.float32(desc.per_level_scale))
    print("wrote", out, len(raw), "->", os.path.getsize(out), "bytes")
