"""One-off parity campaign over DEGENERATE GEOMETRY: triangle soups with zero-area triangles (repeated and collinear vertices), exact
duplicates of triangles (bit-identical t: the tie rule of ray.rs:52-60), coplanar overlapping triangles, many prims with one centroid,
touching and nested spheres, a camera inside a sphere; scenes of one prim, of planes only, without lights; with and without a
transmissive material.  Every frame against the oracle.  usage: python tools/exp/fuzz_campaign4.py FIRST COUNT"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

rtx = importlib.import_module("rust-raytracer_amd")
orc = importlib.import_module("oracle.oracle")
orc.build()
first, count = int(sys.argv[1]), int(sys.argv[2])


def scene(seed):
    rng = np.random.default_rng(seed)
    b = rtx.SceneBuilder()
    kind = int(rng.integers(0, 6))      # 0..2 soup with degenerates, 3 one prim, 4 planes only, 5 spheres
    if rng.random() < 0.85:
        for _ in range(int(rng.integers(1, 4))):
            p = rng.uniform(-8, 8, 3) + np.array([0, 12, 0])
            if rng.random() < 0.4: b.point_light(tuple(p), tuple(rng.uniform(0.2, 0.8, 3)))
            else: b.sphere_light(tuple(p), tuple(rng.uniform(0.2, 0.8, 3)), float(rng.uniform(0.2, 3.0)))
    mats = [b.cook_torrance(k_a=0.05, k_d=0.6, k_s=0.4, k_sg=float(rng.choice([0.0, 0.5, 0.9])), gauss_constant=5.0, roughness=0.1,
                            ior=1.4, diffuse=tuple(rng.uniform(0.2, 1, 3))) for _ in range(3)]
    if rng.random() < 0.4:
        mats.append(b.cook_torrance(k_d=0.1, k_s=0.8, k_sg=0.7, k_tg=0.8, gauss_constant=5.0, roughness=0.02, ior=1.5, diffuse=(1, 1, 1),
                                    transmission=tuple(rng.uniform(0.3, 0.9, 3))))
    mats.append(b.phong(k_a=0.05, k_d=0.7, k_s=0.4, k_sg=0.3, shininess=10.0, ior=1.3, diffuse=(0.7, 0.7, 0.9)))
    pick = lambda: mats[int(rng.integers(0, len(mats)))]
    if kind <= 2:
        n = int(rng.integers(5, 400))
        if rng.random() < 0.5:          # integer coordinates: shared edges, exact coplanarity
            v0 = rng.integers(-6, 7, (n, 3)).astype(np.float64)
            tri = np.stack([v0, v0 + rng.integers(-3, 4, (n, 3)), v0 + rng.integers(-3, 4, (n, 3))], axis=1).astype(np.float64)
        else:
            v0 = rng.uniform(-6, 6, (n, 3))
            tri = np.stack([v0, v0 + rng.normal(0, 2, (n, 3)), v0 + rng.normal(0, 2, (n, 3))], axis=1)
        for i in range(n):
            u = rng.random()
            if u < 0.08: tri[i, 1] = tri[i, 0]                                   # repeated vertex
            elif u < 0.14: tri[i, 2] = tri[i, 0] + 2.5 * (tri[i, 1] - tri[i, 0])  # collinear
            elif u < 0.24 and i: tri[i] = tri[int(rng.integers(0, i))]           # exact duplicate of an earlier triangle
            elif u < 0.28 and i: tri[i] = tri[int(rng.integers(0, i))][[1, 2, 0]]  # the same triangle, vertices rotated
            elif u < 0.32: tri[i] = tri[i].mean(axis=0) + 0.0 * tri[i]            # a point
        if rng.random() < 0.3:           # many prims with one centroid
            c = rng.uniform(-3, 3, 3)
            m = int(rng.integers(20, 60))
            d1, d2 = rng.normal(0, 1, (m, 3)), rng.normal(0, 1, (m, 3))
            extra = np.stack([c + d1, c + d2, c - d1 - d2], axis=1)
            tri = np.concatenate([tri, extra])
        order = rng.permutation(len(tri))
        cut = len(tri) // 2
        b.triangles(tri[order[:cut]].reshape(-1, 9), pick())
        if rng.random() < 0.5: b.sphere(tuple(rng.uniform(-4, 4, 3)), float(rng.uniform(0.5, 3)), pick())
        b.triangles(tri[order[cut:]].reshape(-1, 9), pick())
        if rng.random() < 0.6: b.plane(0, 1, 0, 7, pick())
    elif kind == 3:
        if rng.random() < 0.5: b.triangles(rng.uniform(-4, 4, (1, 9)), pick())
        else: b.sphere((0.5, 0.2, -0.3), 2.0, pick())
    elif kind == 4:
        for _ in range(int(rng.integers(1, 5))):
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            b.plane(float(nrm[0]), float(nrm[1]), float(nrm[2]), float(rng.uniform(2, 9)), pick())
    else:
        c = rng.uniform(-2, 2, 3)
        b.sphere(tuple(c), 3.0, pick())
        b.sphere(tuple(c), 1.5, pick())                                          # nested, concentric
        b.sphere(tuple(c + np.array([4.5, 0, 0])), 1.5, pick())                  # touching the first
        b.sphere(tuple(c + np.array([0, 0, 3.0])), 3.0, pick())                  # overlapping
        if rng.random() < 0.5: b.plane(0, 1, 0, 6, pick())
    b.background = tuple(rng.uniform(0, 1, 3))
    inside = kind == 5 and rng.random() < 0.4
    pos = (c + np.array([0.3, 0.2, 0.1])) if inside else rng.uniform(-3, 3, 3) + np.array([0.0, 3.0, 16.0])
    cam = rtx.camera(tuple(pos), tuple(rng.uniform(-2, 2, 3)), (0, 1, 0), 35.0, 72, 48)
    return b.build(), cam, kind


findings = []
frames = 0
t0 = time.time()
for seed in range(first, first + count):
    flat, cam, kind = scene(seed)
    rng = np.random.default_rng(seed + 99991)
    try:
        sc = rtx.Scene(flat)
    except Exception as e:   # noqa: BLE001
        findings.append("seed %d kind %d: scene refused: %s" % (seed, kind, str(e)[:160])); print("  FINDING " + findings[-1], flush=True); continue
    osc = orc.Scene(flat)
    for trial in range(2):
        kw = dict(reflect_depth=int(rng.integers(1, 5)), refract_depth=int(rng.integers(1, 5)), shadow_samples=int(rng.choice([1, 4, 16])),
                  pixel_samples=int(rng.integers(1, 3)), rng_mode=int(rng.integers(0, 2)), seed=int(rng.integers(0, 100)))
        what = "seed %d kind %d %s" % (seed, kind, kw)
        try:
            rgba, ids, st = sc.render(cam, rtx.options(**kw))
        except Exception as e:   # noqa: BLE001
            findings.append(what + ": " + str(e)[:200]); print("  FINDING " + findings[-1], flush=True); continue
        orgba, oids, ocn = osc.render(cam, rtx.options(**kw), threads=16)
        frames += 1
        n_id = int((ids != oids).sum())
        n_rgb = int((np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) > 1).sum())
        if n_id or n_rgb or int(st.rays_total) != int(ocn.rays_total):
            findings.append("%s: ids %d rgb %d rays %+d" % (what, n_id, n_rgb, int(st.rays_total) - int(ocn.rays_total)))
            print("  FINDING " + findings[-1], flush=True)
    sc.close()
    if (seed - first) % 20 == 19:
        print("  seed %d done, %.0f s, %d frames, %d findings" % (seed, time.time() - t0, frames, len(findings)), flush=True)
print("seeds %d..%d: %d frames of degenerate scenes against the oracle: %d findings" % (first, first + count - 1, frames, len(findings)))
