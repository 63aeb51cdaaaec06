#!/usr/bin/env python3
"""Experiment: one frame as K interleaved shards rendered CONCURRENTLY on one GPU (K scene handles, K host threads,
K streams) against the plain single call.  Hypothesis: while one chain's k_shade waits for registers, the other
chains' shadow launches keep the GPU busy.  usage: exp_two_chains.py K [workload]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("rust-raytracer_amd")
K = int(sys.argv[1])
name, fov, kw = {"sponza": ("sponza", 67.5, dict(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=4)),
                 "bunny": ("bunny", 45.0, dict(reflect_depth=4, refract_depth=4, shadow_samples=1, pixel_samples=2))}[sys.argv[2] if len(sys.argv) > 2 else "sponza"]
named = pkg.host.NamedScene(name)
cam = named.camera(1920, 1080, fov)
scenes = [pkg.Scene(named) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
bufs = [torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda") for _ in range(K)]
opts = [pkg.options(rng_mode=1, seed=0, shard_index=k, shard_count=K, **kw) for k in range(K)]

def frame():
    def work(k):
        scenes[k].render_device(cam, opts[k], bufs[k].data_ptr(), None, streams[k].cuda_stream, want_stats=False)
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()

frame()
t0 = time.perf_counter()
for _ in range(3): frame()
print("K=%d %s: %.2f ms per frame" % (K, name, (time.perf_counter() - t0) / 3 * 1e3))
