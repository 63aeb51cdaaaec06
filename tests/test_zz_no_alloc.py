"""rtx_render_device_noalloc (include/rtx.h): the frame of rtx_render_device out of the library's static device scratch.

The entry point has no counterpart in the reference; it exists for callers that cannot allocate.  It must give the frame
of the existing entry point: primary prim ids bit for bit, ray counts equal, RGBA8 equal up to the order of the f64 atomic
adds into a sample's accumulator (the same +-1 in at most 0.1 % of the pixels that the serial / overlapped test allows).
Two calls in a row share the scratch, the second with a smaller input: whatever the first one left behind must not leak."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _both(rtx, torch, sc, cam, opt):
    H, W = cam.image_height, cam.image_width
    out = {}
    for name in ("render_device", "render_device_noalloc"):
        rgba = torch.full((H, W, 4), 7, dtype=torch.uint8, device="cuda")
        ids = torch.full((H, W), -2, dtype=torch.int32, device="cuda")
        st = getattr(sc, name)(cam, opt, rgba.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[name] = (rgba.cpu().numpy(), ids.cpu().numpy(), st)
    (a, ia, sa), (b, ib, sb) = out["render_device"], out["render_device_noalloc"]
    assert np.array_equal(ia, ib)
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3
    assert (b[..., 3] == 255).all()
    for f in ("rays_primary", "rays_reflect", "rays_gloss", "rays_refract", "rays_shadow"):
        assert getattr(sa, f) == getattr(sb, f), f
    return sa, sb


def test_noalloc_entry_point_matches_render_device(rtx):
    import torch
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32)
    sc = rtx.Scene(flat)
    # first call: 16 shadow tries (cooperative kernel), 4 spp, several wavefront steps out of a 32 MiB workspace
    big = rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=2, rng_mode=1, seed=5)
    sa, sb = _both(rtx, torch, sc, cam_fn(160, 96), big)
    assert sb.rays_shadow > 0 and sb.rays_reflect > 0
    # second call, smaller input, other kernels (per-lane shadows, 1 spp), same scratch
    small = rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=1, pixel_samples=1, rng_mode=1, seed=9)
    _both(rtx, torch, sc, cam_fn(64, 40), small)
    # 4 tries: groups of neighbouring hit points in the cooperative kernel, its left-over list and the per-lane kernel, all in the scratch
    _both(rtx, torch, sc, cam_fn(128, 80), rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=4, pixel_samples=2, rng_mode=1, seed=7))
    sc.close()
    # another handle (transmissive materials: the product walk, two child rays per hit) on the same GPU's scratch
    flat2, cam_fn2 = scenes.mesh_scene(rtx, n_u=24, n_v=16, transmissive=True, point_light=True)
    sc2 = rtx.Scene(flat2)
    _both(rtx, torch, sc2, cam_fn2(96, 64), rtx.options(reflect_depth=4, refract_depth=4, shadow_samples=2, pixel_samples=1, rng_mode=1, seed=3))
    sc2.close()


def test_noalloc_entry_point_allocates_nothing(rtx):
    """What the entry point exists for: on a FRESH scene handle (no workspace, no aux streams, no events yet: rtx_render_device has
    never run on it) a call must leave the device's free memory exactly as it found it.  A regression that routes the call
    through ensure_work() (ray pool, shadow queues: tens of MB) or creates streams / events would move it."""
    import torch
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=32, n_v=24)
    cam = cam_fn(128, 80)
    rgba = torch.zeros((80, 128, 4), dtype=torch.uint8, device="cuda")
    ids = torch.zeros((80, 128), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sc = rtx.Scene(flat)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for opt in (rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=2, rng_mode=1, seed=1),
                rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=4, pixel_samples=1, rng_mode=1, seed=2),
                rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=1, pixel_samples=1, rng_mode=1, seed=3)):
        st = sc.render_device_noalloc(cam, opt, rgba.data_ptr(), ids.data_ptr(), stream)
        assert st.rays_primary > 0
        assert st.ms_total == 0.0 and st.ms_device == 0.0      # rtx.h: the timing fields stay 0
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, (free0, free1)
    # the same handle's first rtx_render_device DOES allocate its workspace: the probe sees allocations of this size
    sc.render_device(cam, rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=1, pixel_samples=1, rng_mode=1, seed=3), rgba.data_ptr(), ids.data_ptr(), stream)
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free2 < free0
    sc.close()


def test_noalloc_entry_point_rejects_what_does_not_fit(rtx):
    import torch
    flat, cam_fn = scenes.sphere_scene(rtx)
    sc = rtx.Scene(flat)
    cam = cam_fn(1920, 1080)
    rgba = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda")
    with pytest.raises(rtx.abi.RtxError) as err:   # 2 M samples x 24 B of accumulators alone exceed the 32 MiB scratch
        sc.render_device_noalloc(cam, rtx.options(2, 2, 1, 1, 1), rgba.data_ptr())
    assert err.value.status == rtx.abi.RTX_ERR_LIMIT and "shard" in str(err.value)
    # ... and the same frame goes through in pieces (8-row blocks dealt over 4 calls), equal to the one-call frame
    ref = torch.zeros_like(rgba)
    opt = dict(reflect_depth=2, refract_depth=2, shadow_samples=1, pixel_samples=1, rng_mode=1, seed=2)
    sc.render_device(cam, rtx.options(**opt), ref.data_ptr())
    for k in range(4):
        sc.render_device_noalloc(cam, rtx.options(shard_index=k, shard_count=4, **opt), rgba.data_ptr())
    torch.cuda.synchronize()
    a, b = ref.cpu().numpy(), rgba.cpu().numpy()
    sc.close()
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3
