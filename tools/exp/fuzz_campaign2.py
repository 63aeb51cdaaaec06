"""One-off parity campaign over OPTIONS: random scenes (tests/test_gpu_fuzz.py; every third with glossy materials and a skybox-less
background) rendered with random depths (0..5 each), shadow try counts (0, 1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 48, 64), pixel and gloss sample
grids, launch sizes (batch_rays from 256 up: hundreds of wavefront steps), walk flags, and through rtx_render, rtx_render_device with compact
shards and rtx_render_device_noalloc; every frame against the oracle.  usage: python tools/exp/fuzz_campaign2.py FIRST COUNT"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import random_scene  # noqa: E402
import torch  # noqa: E402

rtx = importlib.import_module("rust-raytracer_amd")
orc = importlib.import_module("oracle.oracle")
orc.build()
A = rtx.abi
first, count = int(sys.argv[1]), int(sys.argv[2])
findings = []
frames = 0
t0 = time.time()
stream = torch.cuda.current_stream().cuda_stream
for seed in range(first, first + count):
    rng = np.random.default_rng(seed * 7 + 1)
    flat, cam = random_scene(rtx, seed, transmissive=seed % 3 == 2)
    sc = rtx.Scene(flat)
    osc = orc.Scene(flat)
    W, H = cam.image_width, cam.image_height
    for trial in range(3):
        kw = dict(reflect_depth=int(rng.integers(0, 6)), refract_depth=int(rng.integers(0, 6)),
                  shadow_samples=int(rng.choice([0, 1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 48, 64])), gloss_samples=int(rng.integers(1, 4)),
                  pixel_samples=int(rng.integers(1, 4)), rng_mode=int(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)), frame=int(rng.integers(0, 5)))
        batch = int(rng.choice([0, 0, 256, 1024, 4096, 65536]))
        flags = int(rng.choice([0, 0, A.FLAG_NO_COOP, A.FLAG_SERIAL, A.FLAG_OCTREE_WALK, A.FLAG_REFERENCE_ORDER, A.FLAG_SERIAL | A.FLAG_NO_COOP]))
        how = int(rng.integers(0, 3))      # 0 rtx_render, 1 compact shards through rtx_render_device, 2 rtx_render_device_noalloc
        opt_o = rtx.options(**kw)
        orgba, oids, ocn = osc.render(cam, opt_o, threads=16)
        what = "seed %d %s batch %d flags %d how %d" % (seed, kw, batch, flags, how)
        try:
            if how == 0:
                rgba, ids, st = sc.render(cam, rtx.options(batch_rays=batch, flags=flags, **kw))
                rays = int(st.rays_total)
            elif how == 1:
                world = int(rng.integers(2, 5))
                rgba = np.zeros((H, W, 4), np.uint8); ids = np.zeros((H, W), np.int32); rays = 0
                for k in range(world):
                    g = rtx.sharding.StillGather(W, H, k, world, torch.device("cuda", 0))
                    idt = torch.zeros((g.max_blocks * 8, W), dtype=torch.int32, device="cuda")
                    st = sc.render_device(cam, rtx.options(batch_rays=batch, flags=flags | A.FLAG_COMPACT_OUTPUT, shard_index=k, shard_count=world, **kw),
                                          g.target.data_ptr(), idt.data_ptr(), stream)
                    torch.cuda.synchronize()
                    rays += int(st.rays_total)
                    t_, i_ = g.target.cpu().numpy(), idt.cpu().numpy()
                    for j, b in enumerate(rtx.sharding.blocks_of_rank(H, k, world)):
                        rgba[b * 8:b * 8 + 8] = t_[j * 8:j * 8 + 8]; ids[b * 8:b * 8 + 8] = i_[j * 8:j * 8 + 8]
            else:
                out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"); idt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
                st = sc.render_device_noalloc(cam, rtx.options(batch_rays=batch, flags=flags & ~A.FLAG_SERIAL, **kw), out.data_ptr(), idt.data_ptr(), stream)
                torch.cuda.synchronize()
                rgba, ids, rays = out.cpu().numpy(), idt.cpu().numpy(), int(st.rays_total)
        except Exception as e:   # noqa: BLE001
            msg = str(e)
            if how == 2 and "static device scratch" in msg:
                continue             # a documented limit of the allocation-free call, not a finding
            findings.append(what + ": " + msg[:200])
            print("  FINDING " + findings[-1], flush=True)
            continue
        frames += 1
        n_id = int((ids != oids).sum())
        n_rgb = int((np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) > 1).sum())
        if n_id or n_rgb or rays != int(ocn.rays_total):
            findings.append("%s: ids %d rgb %d rays %+d" % (what, n_id, n_rgb, rays - int(ocn.rays_total)))
            print("  FINDING " + findings[-1], flush=True)
    sc.close()
    if (seed - first) % 10 == 9:
        print("  seed %d done, %.0f s, %d frames, %d findings" % (seed, time.time() - t0, frames, len(findings)), flush=True)
print("seeds %d..%d: %d frames with random options, entry points and launch sizes against the oracle: %d findings" % (first, first + count - 1, frames, len(findings)))
for f in findings[:60]:
    print("  " + f)
