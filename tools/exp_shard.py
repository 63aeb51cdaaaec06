import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
rtx = importlib.import_module("rust-raytracer_amd")
import scenes
flat, cam_fn = scenes.mesh_scene(rtx)
cam = cam_fn(96, 64)
sc = rtx.Scene(flat)
for flags in (0, 2):
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=2, pixel_samples=2, rng_mode=1, seed=3, flags=flags)
    full, _, st = sc.render(cam, rtx.options(**kw))
    full2, _, st = sc.render(cam, rtx.options(**kw))
    out = np.zeros_like(full)
    for k in range(3):
        sc.render(cam, rtx.options(shard_index=k, shard_count=3, **kw), want_ids=False, out=out)
    d = np.abs(out.astype(int) - full.astype(int))
    print("flags", flags, "repeat diff", int((full != full2).sum()), "shard diff px", int((d.max(axis=-1) > 0).sum()), "max", d.max())
    if flags == 0: coop = full
    else: print("coop vs nocoop diff px", int((np.abs(coop.astype(int)-full.astype(int)).max(axis=-1) > 0).sum()), np.abs(coop.astype(int)-full.astype(int)).max())
