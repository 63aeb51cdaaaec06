"""One-off parity campaign over the SCENE REGISTRY (my_scene/mod.rs:28-44 through the C++ mirror): all 14 names at random tessellation
(detail 0.08 .. 0.5), frame sizes, fields of view, animation times (Animator::lerp_camera) and options, against the oracle.
usage: python tools/exp/fuzz_campaign5.py FIRST COUNT"""
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
NAMES = "bunny cornell cow easing fresnel heptoroid-shiny heptoroid-refractive heptoroid-white lucy sibenik sphere sponza tachikoma teapot".split()
findings = []
frames = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    name = NAMES[seed % len(NAMES)]
    detail = float(rng.uniform(0.08, 0.5))
    named = rtx.host.NamedScene(name, detail=detail)
    W, H = int(rng.integers(5, 14)) * 8, int(rng.integers(4, 10)) * 8
    t = float(rng.uniform(0.0, 10.0)) if rng.random() < 0.6 else None
    fov = float(rng.uniform(20.0, 70.0))
    try:
        cam = named.camera(W, H, fov, time=t)
    except RuntimeError:            # "Cannot lerp a camera with no keyframes!" (camera.rs: the reference panics): the still camera
        t = None
        cam = named.camera(W, H, fov)
    deep = name.startswith("heptoroid") or name in ("fresnel", "cornell")
    kw = dict(reflect_depth=int(rng.integers(1, 4 if deep else 6)), refract_depth=int(rng.integers(1, 4 if deep else 6)),
              shadow_samples=int(rng.choice([1, 2, 4, 16])), gloss_samples=int(rng.integers(1, 3)), pixel_samples=int(rng.integers(1, 3)),
              rng_mode=int(rng.integers(0, 2)), seed=int(rng.integers(0, 100)), frame=int(rng.integers(0, 200)))
    what = "seed %d %s detail %.2f %dx%d t %s %s" % (seed, name, detail, W, H, t, kw)
    try:
        sc = rtx.Scene(named)
        rgba, ids, st = sc.render(cam, rtx.options(**kw))
        sc.close()
    except Exception as e:   # noqa: BLE001
        findings.append(what + ": " + str(e)[:200]); print("  FINDING " + findings[-1], flush=True); continue
    orgba, oids, ocn = orc.Scene(named).render(cam, rtx.options(**kw), threads=16)
    frames += 1
    n_id = int((ids != oids).sum())
    n_rgb = int((np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) > 1).sum())
    if n_id or n_rgb or int(st.rays_total) != int(ocn.rays_total):
        findings.append("%s: ids %d rgb %d rays %+d" % (what, n_id, n_rgb, int(st.rays_total) - int(ocn.rays_total)))
        print("  FINDING " + findings[-1], flush=True)
    if (seed - first) % 14 == 13:
        print("  seed %d done, %.0f s, %d frames, %d findings" % (seed, time.time() - t0, frames, len(findings)), flush=True)
print("seeds %d..%d: %d frames of the 14 registry scenes at random detail, size, field of view, animation time and options against the oracle: %d findings"
      % (first, first + count - 1, frames, len(findings)))
