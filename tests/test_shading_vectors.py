"""Per-function vectors of the shading half of the path (tests/golden/shading_vectors.json, hex floats, written by
tests/golden/make_shading_vectors.py from the oracle).

CPU: the oracle still reproduces them.  GPU: rtx_shade_probe evaluates the device functions k_shade calls on the same argument
rows.  Bars, per function:
  fresnel, refract, reflect, bbox, texture (checker / uv / image bilinear), skybox: +, -, *, /, sqrt, fmod, floor only
      -> bit for bit (IEEE f64, same operation order, no fused multiply-add on either side);
  Material::sample: Cook-Torrance calls pow, acos, exp and Phong calls pow, whose results differ by ulps between glibc and
      the ROCm device library (both within 1-2 ulp of the true value).  exp(-acos(n.h) / sqrt(roughness)) amplifies an ulp of acos
      by the exponent (up to ~50 for the roughness values used), pow(n.h, shininess) is a single call: relative 1e-12 covers both,
      absolute 1e-15 covers results that cancel to ~0.  NaN / inf results (acos of a dot product above 1, division by n.l = 0) must
      be NaN / the same inf on both sides."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
VEC = json.load(open(os.path.join(HERE, "golden", "shading_vectors.json")))
EXACT = ("fresnel", "refract", "reflect", "bbox", "texture", "skybox")


def _unhex(rows):
    return np.array([[float.fromhex(x) for x in row] for row in rows], dtype=np.float64)


def _check(name, got, want):
    assert got.shape == want.shape
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), name
    g, w = np.where(nan_g, 0.0, got), np.where(nan_w, 0.0, want)
    if name in EXACT:
        bad = np.nonzero(g.view(np.uint64) != w.view(np.uint64))
        assert bad[0].size == 0, "%s: %d values differ, first row %d: %r vs %r" % (name, bad[0].size, bad[0][0], got[bad[0][0]], want[bad[0][0]])
    else:
        inf = np.isinf(w)
        assert np.array_equal(g[inf], w[inf]), name
        g, w = np.where(inf, 0.0, g), np.where(inf, 0.0, w)
        assert np.all(np.abs(g - w) <= 1e-12 * np.abs(w) + 1e-15), "%s: max rel %g" % (name, np.max(np.abs(g - w) / (np.abs(w) + 1e-300)))


@pytest.mark.parametrize("name", sorted(VEC))
def test_oracle_reproduces_shading_vectors(rtx, orc, name):
    mk = importlib.import_module("make_shading_vectors")
    g = VEC[name]
    osc = orc.Scene(mk.SCENES[g["scene"]]())
    got = np.array([mk.evaluate(name, g["probe"], osc, row) for row in _unhex(g["in"])])
    _check(name, got, _unhex(g["out"]))
    assert [[float.hex(float(x)) for x in r] for r in _unhex(g["in"])] == g["in"]     # the file holds what the generator makes


def test_generator_inputs_are_the_committed_ones():
    mk = importlib.import_module("make_shading_vectors")
    for name, (probe, scene, rows) in mk.inputs().items():
        assert VEC[name]["probe"] == probe and VEC[name]["scene"] == scene
        assert [[float.hex(float(x)) for x in r] for r in rows] == VEC[name]["in"], name


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VEC))
def test_device_functions_match_shading_vectors(rtx, name):
    mk = importlib.import_module("make_shading_vectors")
    g = VEC[name]
    sc = rtx.Scene(mk.SCENES[g["scene"]]())
    got = sc.shade_probe(getattr(rtx.abi, "PROBE_" + g["probe"]), _unhex(g["in"]).tolist())
    sc.close()
    want = _unhex(g["out"])
    _check(name, got, want)
    assert np.isfinite(want).mean() > 0.9
