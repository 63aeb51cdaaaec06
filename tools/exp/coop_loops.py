# per-wave loop counters of k_shadow_coop (build: make variant NAME=librtx_loops.so EXTRA=-DRTX_COUNT_LOOPS; run with RTX_LIB=librtx_loops.so)
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-raytracer_amd")
named = pkg.host.NamedScene("sponza")
W, H = 960, 544
cam = named.camera(W, H, 45.0)
sc = pkg.Scene(named)
opt = pkg.options(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=2, rng_mode=1, seed=0, flags=4)
rgba, ids, st = sc.render(cam, opt, want_ids=False)
waves = st.rays_shadow / 64.0
steps, drains, calls = (int(st.work[1][k]) for k in range(3))
lanes, gsteps, devents = (int(st.work[0][k]) for k in range(3))
print("waves %.0f steps/wave %.2f group-steps/wave %.2f (of %.2f) drain events/wave %.2f drain rounds/wave %.2f exact rounds/wave %.2f pair lanes/wave %.1f (%.1f per round)" %
      (waves, steps / waves, gsteps / waves, 4 * steps / waves, devents / waves, drains / waves, calls / waves, lanes / waves, lanes / max(1, calls)))
