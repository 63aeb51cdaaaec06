# the same frame rendered several times under the default overlap: images must agree (f64 atomics in a different order: at most
# one 8-bit step on a few pixels) and ray counts must be equal
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-raytracer_amd")
for name, w, h, kw in (("sponza", 960, 544, dict(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=2)),
                       ("bunny", 960, 544, dict(reflect_depth=4, refract_depth=4, shadow_samples=1, pixel_samples=2)),
                       ("sibenik", 960, 544, dict(reflect_depth=3, refract_depth=3, shadow_samples=4, pixel_samples=2)),
                       ("heptoroid-refractive", 240, 136, dict(reflect_depth=8, refract_depth=8, shadow_samples=1, pixel_samples=1))):
    named = pkg.host.NamedScene(name)
    cam = named.camera(w, h, 45.0)
    sc = pkg.Scene(named)
    opt = pkg.options(rng_mode=1, seed=3, **kw)
    ref = None
    worst = 0
    for i in range(6):
        rgba, ids, st = sc.render(cam, opt, want_ids=False)
        if ref is None:
            ref, rays = rgba.copy(), st.rays_total
        else:
            d = np.abs(rgba.astype(int) - ref.astype(int))
            worst = max(worst, int(d.max()))
            assert st.rays_total == rays, (name, st.rays_total, rays)
            assert d.max() <= 1 and (d > 0).mean() < 1e-3, (name, i, int(d.max()), float((d > 0).mean()))
    print(name, "6 renders agree; largest 8-bit difference", worst, "rays", rays)
    sc.close()
print("OK")
