"""One-off parity campaign beyond the suite's 24 seeds: tests/test_gpu_fuzz.py's random scenes for many more seeds, every option set,
against the oracle; exact counts of pixels whose prim id or colour differs.  usage: python tools/exp/fuzz_campaign.py FIRST COUNT"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import OPTIONS, random_scene  # noqa: E402

rtx = importlib.import_module("rust-raytracer_amd")
orc = importlib.import_module("oracle.oracle")
orc.build()
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
tot_px = id_diff = rgb_diff = rays_diff = t_diff = pid_diff = 0
t0 = time.time()
for seed in range(first, first + count):
    flat, cam = random_scene(rtx, seed, transmissive=seed % 3 == 2)
    sc = rtx.Scene(flat)
    osc = orc.Scene(flat)
    for kw in OPTIONS:
        opt = rtx.options(**kw)
        try:
            rgba, ids, st = sc.render(cam, opt)
        except Exception as e:   # noqa: BLE001
            print("  seed %d options %s: %s" % (seed, kw, e), flush=True)
            bad.append((seed, kw["shadow_samples"], kw["rng_mode"], -1, -1, 0))
            continue
        orgba, oids, ocn = osc.render(cam, opt, threads=16)
        n_id = int((ids != oids).sum())
        n_rgb = int((np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) > 1).sum())
        tot_px += ids.size
        id_diff += n_id
        rgb_diff += n_rgb
        if st.rays_total != ocn.rays_total:
            rays_diff += 1
        if n_id or n_rgb or st.rays_total != ocn.rays_total:
            bad.append((seed, kw["shadow_samples"], kw["rng_mode"], n_id, n_rgb, int(st.rays_total) - int(ocn.rays_total)))
    rng = np.random.default_rng(seed)
    n = 4000
    o = rng.uniform(-140, 140, (n, 3))
    o[n // 2:] *= rng.uniform(3.0, 40.0, (n - n // 2, 1))
    d = rng.uniform(-110, 110, (n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gi, gt = sc.nearest_hits(o, d)
    oi, ot = osc.nearest_hits(o, d)
    t_diff += int((gt.view(np.uint64) != ot.view(np.uint64)).sum())
    pid_diff += int((gi != oi).sum())
    sc.close()
    if (seed - first) % 10 == 9:
        print("  seed %d done, %.0f s, %d findings so far" % (seed, time.time() - t0, len(bad)), flush=True)
print("seeds %d..%d: %d pixels x 4 option sets: prim-id differences %d, pixels off by more than one 8-bit step %d, frames with another ray count %d; "
      "%d random rays: t differs on %d, prim on %d" % (first, first + count - 1, tot_px // 4, id_diff, rgb_diff, rays_diff, count * 4000, t_diff, pid_diff))
for b in bad[:40]:
    print("  seed %d shadow_samples %d rng %d: ids %d rgb %d rays %+d" % b)
