#!/usr/bin/env python3
"""Per-function golden vectors for the shading half of the path, as hex floats:
tests/golden/shading_vectors.json.  Generated with the CPU oracle (oracle/oracle.cpp); the reference holds no test for any of these
functions (fresnel_reflect, refract, reflect, Material::sample, Texture::color, CubeMap::color), so like the other golden files they
pin the ORACLE's restatement; BBox::intersects is pinned by the reference's own test as well (tests/test_oracle_known_answers.py).

    python tests/golden/make_shading_vectors.py

Argument rows use the layout of rtx_shade_probe (include/rtx.h); outputs are 4 columns (unused ones 0)."""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402

rtx = importlib.import_module("rust-raytracer_amd")
orc = importlib.import_module("oracle.oracle")

SCENES = {"kitchen": lambda: scenes.kitchen_sink_scene(rtx)[0], "mesh": lambda: scenes.mesh_scene(rtx)[0]}


def unit(v):
    return v / np.linalg.norm(v)


def inputs():
    """name -> (probe, scene, rows)"""
    rng = np.random.RandomState(2024)
    g = {}
    iors = [0.05, 0.3, 0.98, 1.0, 1.33, 1.5, 3.0]
    rows = []
    for k in range(256):
        n = unit(rng.normal(size=3))
        i = unit(rng.normal(size=3))
        if k % 4:          # mostly on the side of the normal the renderer sees
            i = i if np.dot(i, n) > 0 else -i
        rows.append([iors[k % 7], *i, *n, float(k % 2)])
    g["fresnel"] = ("FRESNEL", "kitchen", rows)
    rows = []
    for k in range(256):
        rows.append([*unit(rng.normal(size=3)), *unit(rng.normal(size=3)), iors[k % 7], float((k // 7) % 2)])
    g["refract"] = ("REFRACT", "kitchen", rows)
    g["reflect"] = ("REFLECT", "kitchen", [[*unit(rng.normal(size=3)), *rng.normal(size=3)] for _ in range(64)])
    rows = []
    for k in range(256):
        lo = rng.uniform(-2, 1, size=3)
        hi = lo + rng.uniform(0.0, 2.0, size=3) * (k % 5 != 0)       # every fifth box is flat in all axes (a point)
        o = rng.uniform(-3, 3, size=3)
        d = unit(rng.normal(size=3))
        if k % 3 == 0:                                               # axis-parallel: zero components, inf inverse
            ax = (k // 3) % 3
            d = np.eye(3)[ax] * (1.0 if k % 2 else -1.0)                # all three axes, both signs
            if (k // 6) % 2:                                            # half of them aimed at the box's cross-section: hits as well as misses
                for b in range(3):
                    if b != ax:
                        o[b] = lo[b] + (hi[b] - lo[b]) * (0.25 + 0.5 * ((k // 12 + b) % 2))
        if k % 7 == 0:
            o[(k // 7) % 3] = lo[(k // 7) % 3]                       # origin on a face plane (0 * inf = NaN with a zero component), any axis
        rows.append([*lo, *hi, *o, *d])
    g["bbox"] = ("BBOX", "kitchen", rows)
    # kitchen sink: textures 0 image, 1 uv, 2 checker; materials 0..5 (CT+image, Phong+uv, Phong+checker, Flat, CT mirror, CT glass)
    rows = []
    for k in range(384):
        u, v = rng.uniform(-3, 3, size=2)
        if k % 16 == 0:
            u, v = float(k % 5) * 0.125, float(k % 3) * 0.25          # on checker / texel boundaries
        rows.append([float(k % 3), u, v])
    g["texture"] = ("TEXTURE", "kitchen", rows)
    rows = []
    for k in range(384):
        n = unit(rng.normal(size=3))
        i = unit(rng.normal(size=3))
        l = unit(rng.normal(size=3))
        if k % 8:                                                    # front-facing, lit: the common case
            i = i if np.dot(i, n) > 0 else -i
            l = l if np.dot(l, n) > 0 else -l
        rows.append([float(k % 6), *n, *i, *l, *rng.uniform(-2, 2, size=2)])
    g["material"] = ("MATERIAL", "kitchen", rows)
    rows = [list(unit(rng.normal(size=3))) for _ in range(256)]
    rows += [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [1, -1, 1], [-1, 1, 1], [0.5, 0.5, -0.5]]
    g["skybox"] = ("SKYBOX", "mesh", [list(map(float, r)) for r in rows])
    return g


def evaluate(name, probe, osc, row):
    """One row through the oracle; 4 output columns like rtx_shade_probe."""
    a = np.array(row, dtype=np.float64)
    if probe == "FRESNEL":
        return [orc.fresnel_reflect(a[0], a[1:4], a[4:7], bool(a[7])), 0.0, 0.0, 0.0]
    if probe == "REFRACT":
        t = orc.refract(a[0:3], a[3:6], a[6], bool(a[7]))
        return [0.0, 0.0, 0.0, 0.0] if t is None else [*t, 1.0]
    if probe == "REFLECT":
        return [*orc.reflect(a[0:3], a[3:6]), 0.0]
    if probe == "BBOX":
        return [1.0 if orc.bbox_intersects(a[0:3], a[3:6], a[6:9], a[9:12]) else 0.0, 0.0, 0.0, 0.0]
    if probe == "TEXTURE":
        return [*osc.texture_color(int(a[0]), a[1], a[2]), 0.0]
    if probe == "MATERIAL":
        return [*osc.material_sample(int(a[0]), a[1:4], a[4:7], a[7:10], a[10], a[11]), 0.0]
    if probe == "SKYBOX":
        return [*osc.skybox_color(a[0:3]), 0.0]
    raise KeyError(probe)


def main():
    out = {}
    oscs = {k: orc.Scene(f()) for k, f in SCENES.items()}
    for name, (probe, scene, rows) in inputs().items():
        res = [evaluate(name, probe, oscs[scene], r) for r in rows]
        out[name] = {"probe": probe, "scene": scene,
                     "in": [[float.hex(float(x)) for x in r] for r in rows],
                     "out": [[float.hex(float(x)) for x in r] for r in res]}
        print(name, len(rows), "rows")
    with open(os.path.join(HERE, "shading_vectors.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
