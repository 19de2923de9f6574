"""Writes tests/golden/oracle_render_golden.npz: a 64x36 frame of the synthetic lego-like scene with the cage edit, rendered
by the CPU oracle.  The reference itself ships no fixtures for this path and cannot be run, so this golden
pins the ORACLE (against drift), not the reference.   python tests/golden/make_render_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Scene  # noqa: E402

s = Scene(1, True, 6)
s.oracle_model.set_bitfield(s.edited_bitfield)
p = s.params_for(64, 36, 60.0)
f, d, st, stats = s.oracle_model.render(p, [s.oracle_edit])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_render_golden.npz"), frame=f, depth=d, steps=st)
print("hit", stats.n_hit, "samples", stats.composited)
