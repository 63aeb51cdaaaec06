"""One-off: parity on LARGE scenes (the sponza and bunny stand-ins at 4x and 9x their triangle counts: 1.0 M and 2.4 M / 0.63 M
triangles): scene build (device-side hierarchy), a small frame with 16 shadow tries and one deterministic frame against the oracle."""
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
for name, fov, detail in (("sponza", 67.5, 2.0), ("bunny", 45.0, 3.0), ("sponza", 67.5, 3.0)):
    t0 = time.time()
    named = rtx.host.NamedScene(name, detail=detail)
    t1 = time.time()
    sc = rtx.Scene(named)
    info = sc.info()
    t2 = time.time()
    print("%s detail %.1f: %d triangles, generate %.1f s, rtx_scene_create %.2f s, octree depth %d, %d nodes, %d leaf entries, wide nodes %d, bvh4 nodes %d, on device %d"
          % (name, detail, named.n_triangles, t1 - t0, t2 - t1, info.octree_depth, info.n_nodes, info.n_leaf_entries, info.wide_nodes, info.bvh4_nodes,
             info.hierarchy_on_device), flush=True)
    osc = orc.Scene(named)
    print("  oracle scene built in %.1f s" % (time.time() - t2), flush=True)
    cam = named.camera(160, 96, fov)
    for kw in (dict(reflect_depth=3, refract_depth=3, shadow_samples=1, pixel_samples=1, rng_mode=0),
               dict(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=2, rng_mode=1, seed=4)):
        rgba, ids, st = sc.render(cam, rtx.options(**kw))
        t3 = time.time()
        orgba, oids, ocn = osc.render(cam, rtx.options(**kw), threads=16)
        n_id = int((ids != oids).sum())
        n_rgb = int((np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) > 1).sum())
        print("  %s: %d rays (oracle %d, %.1f s), %.2f ms device; prim-id differences %d, pixels off by more than one step %d"
              % (kw, st.rays_total, ocn.rays_total, time.time() - t3, st.ms_device, n_id, n_rgb), flush=True)
    sc.close()
print("OK")
