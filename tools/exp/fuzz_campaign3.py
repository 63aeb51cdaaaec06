"""One-off parity campaign over SCALES, POSITIONS and MATERIAL KINDS: the hand-made test scenes (tests/scenes.py: every material, texture
and light kind, glossy mirrors, glass, planes, spheres, meshes, an axis-aligned scene) under a random similarity transform (scale
10^-3 .. 10^3, translation up to 1000 scales away from the origin: the f32 frames of the fast walks are relative to the scene's cube),
random cameras around them and random options (gloss grids too), against the oracle.  usage: python tools/exp/fuzz_campaign3.py FIRST COUNT"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402

rtx = importlib.import_module("rust-raytracer_amd")
orc = importlib.import_module("oracle.oracle")
orc.build()
first, count = int(sys.argv[1]), int(sys.argv[2])


def make(which):
    if which == 0: return scenes.kitchen_sink_scene(rtx, "mixed")
    if which == 1: return scenes.kitchen_sink_scene(rtx, "points")
    if which == 2: return scenes.cornell_scene(rtx, glossy=True)
    if which == 3: return scenes.cornell_scene(rtx, glossy=False)
    if which == 4: return scenes.mesh_scene(rtx, n_u=40, n_v=28, transmissive=True)
    if which == 5: return scenes.mesh_scene(rtx, n_u=40, n_v=28, third_light=True)
    if which == 6: return scenes.axis_aligned_scene(rtx)
    if which == 7: return scenes.axis_aligned_scene(rtx, transmissive=True)
    if which == 8: return scenes.cube_obj_scene(rtx)
    return scenes.sphere_scene(rtx)


def transform(flat, s, t):
    flat.tri_vertices *= s; flat.tri_vertices += t
    flat.sphere_center *= s; flat.sphere_center += t
    flat.sphere_radius *= s
    for p in flat.plane_abcd:                       # n . x + d = 0  ->  n . x' + (s d - n . t) = 0
        p[3] = s * p[3] - float(np.dot(p[:3], t))
    for i in range(flat.n_lights):
        L = flat.lights[i]
        for a in range(3):
            L.position[a] = s * L.position[a] + t[a]
        L.radius *= s
    for i in range(flat.n_textures):                # checker cells scale with the scene (plane UVs are world coordinates)
        flat.textures[i].scale *= s


findings = []
frames = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    which = int(rng.integers(0, 10))
    flat, cam_fn = make(which)
    s = float(10.0 ** rng.uniform(-3, 3)) if rng.random() < 0.8 else 1.0
    t = rng.uniform(-1000, 1000, 3) * s if rng.random() < 0.7 else np.zeros(3)
    cam0 = cam_fn(80, 56)
    pos = np.array(list(cam0.position)); look = np.array(list(cam0.look_at))
    if rng.random() < 0.6:                          # another viewpoint at about the same distance
        dist = np.linalg.norm(pos - look)
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        v[1] = abs(v[1]) * 0.6 + 0.05
        pos = look + v / np.linalg.norm(v) * dist * rng.uniform(0.5, 1.3)
    transform(flat, s, t)
    cam = rtx.camera(tuple(pos * s + t), tuple(look * s + t), (0, 1, 0), float(cam0.fov_deg), 80, 56)
    sc = rtx.Scene(flat)
    osc = orc.Scene(flat)
    for trial in range(2):
        kw = dict(reflect_depth=int(rng.integers(1, 6)), refract_depth=int(rng.integers(1, 6)),
                  shadow_samples=int(rng.choice([1, 2, 4, 8, 16, 32])), gloss_samples=int(rng.integers(1, 4)),
                  pixel_samples=int(rng.integers(1, 3)), rng_mode=int(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)))
        what = "seed %d scene %d scale %.3g shift %s %s" % (seed, which, s, np.round(t, 1).tolist(), kw)
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
print("seeds %d..%d: %d frames of the hand-made scenes under random scales, shifts, cameras and options against the oracle: %d findings" % (first, first + count - 1, frames, len(findings)))
