"""Packs the output of oracle/_ref/decision_vectors (the reference's own isUnchangeable / solvePlaneEquations /
eigenDecomposition text evaluated on seeded inputs, oracle/ref_decision_vectors.cpp) into
tests/golden/ref_decision_vectors.npz.  Build container only:

    make -C oracle ref && python tests/golden/make_ref_decision_vectors.py
"""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# two generators (round 6): decision_vectors holds no stand-in (guard + plane solve); decision_eigen_vectors alone defines rsqrtf
d = json.loads(subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "decision_vectors")], text=True))
assert not any(k.startswith("eig_") for k in d)
e = json.loads(subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "decision_eigen_vectors")], text=True))
assert all(k.startswith("eig_") for k in e)
d.update(e)
conv = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf}
arrs = {k: np.array([conv[x] if isinstance(x, str) else x for x in v], np.float32) for k, v in d.items()}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_decision_vectors.npz"), **arrs)
print({k: v.shape for k, v in arrs.items()})
