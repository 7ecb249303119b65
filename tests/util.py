"""Shared helpers for the parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def digest(a):
    """sha1 over float32 values with -0.0 folded into +0.0 (same as oracle/make_golden.py)."""
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha1(a.tobytes()).hexdigest()


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def have(name):
    return os.path.exists(os.path.join(GOLDEN, name + ".npz"))


def golden_input(g, so):
    """Regenerate the input volume a fixture was made from and check its digest."""
    spec = json.loads(str(g["input_spec"]))
    if spec["gen"] == "survey":
        vol = so.synth_survey(spec["n"] if not isinstance(spec["n"], list) else tuple(spec["n"]),
                              nblob=spec.get("nblob"))
    else:
        vol = so.synth_lattice(spec["n"], seed=spec["seed"])
    assert digest(vol) == str(g["input_digest"]), "synthetic generator drifted"
    return vol


def golden_params(g):
    return {k[len("param_"):]: g[k].item() for k in g.files if k.startswith("param_")}


def rel_err(a, b):
    """max |a-b| / max(|a|,|b|) over elements where either is non-zero."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.maximum(np.abs(a), np.abs(b))
    m = den > 0
    if not m.any():
        return 0.0
    return float((np.abs(a - b)[m] / den[m]).max())


def desc_projection(h):
    """The projection oracle/make_golden.py stores as `desc_proj` (same seed, same matrix)."""
    P = np.random.default_rng(20240768).standard_normal((768, 8))
    P /= np.linalg.norm(P, axis=0, keepdims=True)
    return np.ascontiguousarray(h, np.float64) @ P


def assert_desc_projection(hist, want_proj, rtol=1e-5):
    """Every row of `hist` (N x 768) against the reference's projected rows: each of the 8
    projections within rtol * |row|_2 (unit projection vectors: an elementwise relative error e
    moves a projection by at most e * |row|_2; mass in a wrong bin moves it by far more)."""
    hist = np.ascontiguousarray(hist, np.float64)
    assert want_proj.shape == (len(hist), 8)
    norm = np.sqrt((hist * hist).sum(axis=1))
    err = np.abs(desc_projection(hist) - want_proj).max(axis=1)
    bad = np.nonzero(err > rtol * norm)[0]
    assert len(bad) == 0, "rows %s: projection error %s of row norm" % (bad[:8], (err / norm)[bad[:8]])
    return float((err / np.maximum(norm, 1e-30)).max())
