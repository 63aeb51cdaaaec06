# how the exact tests of k_shadow_coop end (build: make variant NAME=librtx_class.so EXTRA=-DRTX_COUNT_CLASS; run with RTX_LIB=librtx_class.so)
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-raytracer_amd")
for name, ss, ps in (("sponza", 16, 2), ("sibenik", 4, 2), ("bunny", 16, 1)):
    named = pkg.host.NamedScene(name)
    W, H = 960, 544
    cam = named.camera(W, H, 45.0)
    sc = pkg.Scene(named)
    opt = pkg.options(reflect_depth=3, refract_depth=3, shadow_samples=ss, pixel_samples=ps, rng_mode=1, seed=0, flags=4)
    rgba, ids, st = sc.render(cam, opt, want_ids=False)
    c = [int(st.work[k // 3][k % 3]) for k in range(6)]
    tot = max(1, sum(c))
    print(name, "exact tests per feeler %.2f:" % (tot / st.rays_shadow),
          " ".join("%s %.3f" % (n, v / tot) for n, v in zip(("hit", "bary_out>1e-3", "bary_out<1e-3", "|t|<1e-9", "behind", "beyond"), c)))
