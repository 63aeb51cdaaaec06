"""Child process of tests/test_gpu_variants.py (not collected by pytest): renders small scenes with the library build named by
RTX_LIB (a forced-failure build of rust-raytracer_amd/csrc) and compares them with the CPU oracle.  Prints OK on success."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402

rtx = importlib.import_module("rust-raytracer_amd")
orc = importlib.import_module("oracle.oracle")
orc.build()
assert os.path.basename(rtx.abi.LIB_PATH) == os.environ["RTX_LIB"], rtx.abi.LIB_PATH


def compare(flat, cam, opt, what):
    sc = rtx.Scene(flat)
    rgba, ids, st = sc.render(cam, opt)
    sc.close()
    orgba, oids, ocn = orc.Scene(flat).render(cam, opt)
    id_ok = float((ids == oids).mean())
    rgb_ok = float((np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) <= 1).mean())
    assert st.rays_total == ocn.rays_total, (what, st.rays_total, ocn.rays_total)
    assert id_ok >= 0.9999 and rgb_ok >= 0.999, (what, id_ok, rgb_ok)
    print("%s: ids %.5f rgb %.5f rays %d" % (what, id_ok, rgb_ok, st.rays_total))


flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32)
# closest hit + one feeler per light and lane (the per-lane any-hit walk)
compare(flat, cam_fn(160, 96), rtx.options(reflect_depth=4, refract_depth=4, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0),
        "deterministic, 1 shadow try")
# 16 tries per hit point: the cooperative shadow walk; 4: groups of neighbouring hit points + the per-lane walk for the rest
for tries in (16, 4):
    compare(flat, cam_fn(96, 64), rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=tries, gloss_samples=1, pixel_samples=2,
                                              rng_mode=1, seed=11), "philox, %d shadow tries" % tries)
# a transmissive material: the two-phase shadow product (distinct hit prims over the 4-wide hierarchy, then their multiplicities)
flat_t, cam_fn_t = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=True)
compare(flat_t, cam_fn_t(96, 64), rtx.options(reflect_depth=4, refract_depth=4, shadow_samples=2, gloss_samples=1, pixel_samples=1, rng_mode=1, seed=4),
        "transmissive, 2 shadow tries")
if os.environ["RTX_LIB"] == "librtx_smallstack.so":
    # a frame whose shadow launches hand over more feelers than a retry list holds (2^14 per light in this build): the library notices the flag after the frame and renders it again with the all-f64 shadow walk; scenes
    # with transmissive materials keep that list too (the two-phase shadow product)
    W, H, rows = 768, 512, (224, 288)
    opt = rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=1, gloss_samples=1, pixel_samples=2, rng_mode=1, seed=9)
    sc = rtx.Scene(flat_t)
    rgba, ids, st = sc.render(cam_fn_t(W, H), opt)
    sc.close()
    assert st.exact_shadow_rerenders == 1, st.exact_shadow_rerenders
    orgba, oids, _ = orc.Scene(flat_t).render(cam_fn_t(W, H), opt, threads=16, rows=rows)
    r = slice(*rows)
    assert (ids[r] >= 0).mean() > 0.3
    id_ok = float((ids[r] == oids[r]).mean())
    rgb_ok = float((np.abs(rgba[r][..., :3].astype(int) - orgba[r][..., :3].astype(int)).max(axis=-1) <= 1).mean())
    assert id_ok >= 0.9999 and rgb_ok >= 0.999, ("retry overflow, transmissive", id_ok, rgb_ok)
    print("retry-list overflow in a transmissive scene: re-rendered, ids %.5f rgb %.5f" % (id_ok, rgb_ok))
print("OK")
