import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-raytracer_amd")
named = pkg.host.NamedScene("heptoroid-refractive")
cam = named.camera(240, 136, 45.0)
sc = pkg.Scene(named)
print(sc.info().n_leaf_entries, sc.info().n_nodes, named.n_triangles)
opt = pkg.options(reflect_depth=8, refract_depth=8, shadow_samples=1, pixel_samples=1, rng_mode=1, seed=0, flags=4)
rgba, ids, st = sc.render(cam, opt, want_ids=False)
cl = st.rays_primary + st.rays_reflect + st.rays_refract
print("closest per ray", [st.work[0][k] / cl for k in range(3)], "shadow per ray", [st.work[1][k] / st.rays_shadow for k in range(3)], st.ms_closest, st.ms_shadow)
