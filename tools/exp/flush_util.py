# test-round utilisation of traverse_bvh_dealt (build: make variant NAME=librtx_flush.so EXTRA="-DRTX_COUNT_WORK -DRTX_COUNT_FLUSH").
# Only the dealt walk carries these counters now (scenes with transmissive materials: the heptoroid lines); the figures for
# traverse_bvh quoted in DESIGN.md (18-28 % of the lanes busy in a flush round) were taken with the same counters in that walk
# before the two were separated.
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-raytracer_amd")
for name, w, h, d, ss, ps in (("heptoroid-refractive", 240, 136, 8, 1, 1), ("bunny", 960, 544, 4, 1, 2), ("sibenik", 960, 544, 3, 1, 2), ("sponza", 960, 544, 3, 1, 2)):
    named = pkg.host.NamedScene(name)
    cam = named.camera(w, h, 45.0)
    sc = pkg.Scene(named)
    opt = pkg.options(reflect_depth=d, refract_depth=d, shadow_samples=ss, pixel_samples=ps, rng_mode=1, seed=0, flags=4 | 2)   # serial, no cooperative kernel
    rgba, ids, st = sc.render(cam, opt, want_ids=False)
    cl = st.rays_primary + st.rays_reflect + st.rays_refract
    for tag, n, wk in (("closest", cl, st.work[0]), ("shadow", st.rays_shadow, st.work[1])):
        waves = n / 64.0
        print("%s %s: wave-steps %.1f flush rounds %.1f tests per ray %.2f => lanes busy in a flush round %.2f" %
              (name, tag, wk[0] / waves, wk[1] / waves, wk[2] / n, wk[2] / max(1, wk[1]) / 64.0))
