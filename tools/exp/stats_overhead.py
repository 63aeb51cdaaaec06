"""Frame time with and without the per-kernel timing events (rtx_stats requested or not): stats_overhead.py workload [frames]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
rtx = importlib.import_module("rust-raytracer_amd")
w = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sys.path.insert(0, ROOT)
from bench import WORKLOADS
name, W, H, fov, refl, refr, shadow, gloss, ps = WORKLOADS[w]
named = rtx.host.NamedScene(name)
cam = named.camera(W, H, fov)
sc = rtx.Scene(named)
opt = rtx.options(reflect_depth=refl, refract_depth=refr, shadow_samples=shadow, gloss_samples=gloss, pixel_samples=ps, rng_mode=1, seed=0)
rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for want in (True, False, True, False):
    for _ in range(3): sc.render_device(cam, opt, rgba.data_ptr(), None, stream, want_stats=want)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): sc.render_device(cam, opt, rgba.data_ptr(), None, stream, want_stats=want)
    torch.cuda.synchronize()
    print("%s stats=%s: %.3f ms per frame" % (w, want, (time.perf_counter() - t0) * 1e3 / n), flush=True)
sc.close()
