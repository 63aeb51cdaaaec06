"""Where the time of an animation frame goes outside rtx_render_device (bench.py --animate): per-piece host timings."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("rust-raytracer_amd")
named = pkg.host.NamedScene("sibenik")
scene = pkg.Scene(named, device=0)
W, H = 1920, 1080
rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
T = {"camera": 0.0, "options": 0.0, "render": 0.0, "on_frame": 0.0, "lib_ms_total": 0.0, "lib_ms_device": 0.0}
N = 24
for k in range(N + 2):
    if k == 2:
        for key in T: T[key] = 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
    t0 = time.perf_counter(); cam = named.camera(W, H, 45.0, time=k / 25.0); t1 = time.perf_counter()
    opt = pkg.options(reflect_depth=3, refract_depth=3, shadow_samples=4, gloss_samples=1, pixel_samples=2, rng_mode=pkg.abi.RNG_PHILOX, seed=0, frame=k)
    t2 = time.perf_counter()
    st = scene.render_device(cam, opt, rgba.data_ptr(), None, stream, want_stats=True); t3 = time.perf_counter()
    s = int(rgba[::37, ::41].sum().item()); t4 = time.perf_counter()
    T["camera"] += t1 - t0; T["options"] += t2 - t1; T["render"] += t3 - t2; T["on_frame"] += t4 - t3
    T["lib_ms_total"] += st.ms_total * 1e-3; T["lib_ms_device"] += st.ms_device * 1e-3
torch.cuda.synchronize(); t_all = time.perf_counter() - t_all
print("per frame ms:", {k: round(v / N * 1e3, 3) for k, v in T.items()}, "all", round(t_all / N * 1e3, 3))
