import importlib, sys, time
sys.path.insert(0, "/root/repo")
rtx = importlib.import_module("rust-raytracer_amd")
for detail in (1.0, 3.0):
    named = rtx.host.NamedScene("bunny", detail=detail)
    sc = rtx.Scene(named)
    cam = named.camera(160, 96, 45.0)
    for kw in (dict(shadow_samples=16, pixel_samples=2, rng_mode=1, seed=4), dict(shadow_samples=16, pixel_samples=2, rng_mode=1, seed=4), dict(shadow_samples=16, pixel_samples=2, rng_mode=0),
               dict(shadow_samples=1, pixel_samples=2, rng_mode=1), dict(shadow_samples=16, pixel_samples=2, rng_mode=1, seed=4, flags=rtx.abi.FLAG_SERIAL), dict(shadow_samples=16, pixel_samples=2, rng_mode=1, seed=4, flags=rtx.abi.FLAG_NO_COOP)):
        t0 = time.time()
        rgba, ids, st = sc.render(cam, rtx.options(reflect_depth=3, refract_depth=3, **kw))
        print(detail, kw, "rays %d device %.2f ms closest %.2f shade %.2f shadow %.2f %s launches %d/%d rerender %d wall %.1f ms" % (st.rays_total, st.ms_device, st.ms_closest, st.ms_shade, st.ms_shadow,
              [round(x, 2) for x in list(st.ms_shadow_light)[:2]], st.launches_closest, st.launches_shadow, st.exact_shadow_rerenders, (time.time() - t0) * 1e3), sc.last_kernel_names(), flush=True)
    sc.close()
