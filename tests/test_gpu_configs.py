"""BASELINE.json's configs at the settings they are quoted on, against the CPU oracle (-m gpu).

  C3 sponza  : pixel_samples 4 (16 spp), 16 shadow samples  -> test_gpu_parity.test_baseline_scenes_full_size_properties
  C4 heptoroid-refractive : reflect / refract depth 16 (this file)
  C5 sibenik : pixel_samples 2, 4 shadow samples, and frames of the 0-7 s camera animation (this file)

and the error paths of the wavefront loop: a launch whose retry list overflows, a pool that runs out in mid-frame."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _compare(rgba, ids, orgba, oids, id_bar=0.9999, rgb_bar=0.999):
    id_ok = (ids == oids).mean()
    diff = np.abs(rgba[..., :3].astype(np.int32) - orgba[..., :3].astype(np.int32)).max(axis=-1)
    rgb_ok = (diff <= 1).mean()
    assert id_ok >= id_bar, "prim-id parity %.6f" % id_ok
    assert rgb_ok >= rgb_bar, "rgb +-1 parity %.6f (max diff %d)" % (rgb_ok, diff.max())
    return id_ok, rgb_ok


@pytest.mark.parametrize("detail,size", [(0.05, (32, 64)), (0.07, (8, 64))])
def test_heptoroid_refractive_depth_16(rtx, orc, detail, size):
    """BASELINE config 4 as stated: reflect_depth = refract_depth = 16 (renderer.rs:105-186 recursing 30 levels; k_sg = k_tg = 1,
    ior 1.5, transmission 0.8: every hit spawns a mirror and a transmission child, the shadow product runs over the
    multiset).  The ray tree grows ~x7 per two levels (one hit pixel of the detail-0.07 mesh spawns about 10^6 rays),
    so the frame is tiny and tall (8-row tiles keep the oracle's threads busy).  Pool sizing at levels = 30
    (need_per_batch = 61), the 12-bit depth counters of the ray flags and the segment stack are all in play; a second
    render with a 4096-ray batch pushes the pool to its bound and the segment stack to its full depth."""
    named = rtx.host.NamedScene("heptoroid-refractive", detail=detail)
    cam = named.camera(size[0], size[1], 45.0)
    kw = dict(reflect_depth=16, refract_depth=16, shadow_samples=1, pixel_samples=1, rng_mode=0)
    orgba, oids, ocn = orc.Scene(named).render(cam, rtx.options(**kw), threads=16)
    sc = rtx.Scene(named)
    rgba, ids, st = sc.render(cam, rtx.options(**kw))
    rgba2, ids2, st2 = sc.render(cam, rtx.options(batch_rays=4096, **kw))
    sc.close()
    assert (oids >= 0).sum() >= 5 and ocn.rays[3] > 100000
    assert np.array_equal(ids, oids) and np.array_equal(ids2, oids)
    # a handful of hit pixels, each the sum of ~10^5..10^6 paths: every one of them within +-1
    assert (np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)) <= 1).all()
    assert (np.abs(rgba2[..., :3].astype(int) - orgba[..., :3].astype(int)) <= 1).all()
    for got in (st, st2):   # identical ray accounting: 1 per get_nearest_hit, 1 per shadow try
        assert (got.rays_primary, got.rays_reflect, got.rays_gloss, got.rays_refract, got.rays_shadow) == tuple(int(x) for x in ocn.rays[:5])
    assert st2.launches_closest > st.launches_closest and st2.peak_pool_rays <= 4096 * 61


def test_sibenik_at_config_settings(rtx, orc):
    """BASELINE config 5's still settings (pixel_samples 2, 4 shadow samples, depth 3/3, three lights) at 1920x1080: the
    whole frame through both shadow traversals and, on two 8-row blocks, against the oracle."""
    named = rtx.host.NamedScene("sibenik")
    cam = named.camera(1920, 1080, 45.0)
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=4, pixel_samples=2, rng_mode=1, seed=0)
    sc = rtx.Scene(named)
    full, ids, st = sc.render(cam, rtx.options(**kw))
    alt, ids2, _ = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_NO_COOP, **kw))
    sc.close()
    assert np.array_equal(ids, ids2)
    assert (np.abs(full.astype(int) - alt.astype(int)) <= 1).all() and (full != alt).mean() < 1e-3
    osc = orc.Scene(named)
    for b in (30, 95):
        o_rgba, o_ids, _ = osc.render(cam, rtx.options(**kw), threads=16, rows=(b * 8, b * 8 + 8))
        rows = slice(b * 8, b * 8 + 8)
        assert (ids[rows] == o_ids[rows]).mean() >= 0.9999
        assert (np.abs(full[rows, :, :3].astype(int) - o_rgba[rows, :, :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999


@pytest.mark.parametrize("t", [0.0, 3.5, 7.0])
def test_sibenik_animation_frames(rtx, orc, t):
    """Frames of config 5's 0-7 s camera animation: the camera of time t comes out of the host mirror's Animator::lerp_camera
    (animator.rs:44-48, 91-114: keyframe pair, Bezier easing, lerped position / look_at / up) and the frame
    (frame number = floor(t * 25), part of the RNG key) is compared with the oracle's at that camera."""
    named = rtx.host.NamedScene("sibenik", detail=0.5)
    cam = named.camera(480, 272, 45.0, time=t)
    still = named.camera(480, 272, 45.0)
    if t > 0.0:
        assert list(cam.position) != list(still.position)   # the camera does move
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=4, pixel_samples=2, rng_mode=1, seed=0, frame=int(t * 25))
    orgba, oids, ocn = orc.Scene(named).render(cam, rtx.options(**kw), threads=16)
    sc = rtx.Scene(named)
    rgba, ids, st = sc.render(cam, rtx.options(**kw))
    sc.close()
    _compare(rgba, ids, orgba, oids)
    assert abs(int(st.rays_total) - int(ocn.rays_total)) <= 1e-3 * ocn.rays_total + 2


def _floor_scene(rtx, n=24):
    """A tessellated floor in the plane y = 0 under a POINT light that lies in that plane, and a box above it: every feeler
    from a floor point has a y component of ~1e-16 / distance (the hit point's y is o + d * t rounding noise), i.e. is
    degenerate (< 1e-9) and goes to the launch's retry list."""
    b = rtx.SceneBuilder()
    b.point_light((6.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    grey = b.cook_torrance(k_a=0.1, k_d=0.8, k_s=0.2, gauss_constant=5.0, roughness=0.2, ior=1.5, diffuse=(0.8, 0.8, 0.8))
    xs = np.linspace(-8.0, 8.0, n + 1)
    v = []
    for i in range(n):
        for j in range(n):
            p00, p10 = (xs[i], 0.0, xs[j]), (xs[i + 1], 0.0, xs[j])
            p01, p11 = (xs[i], 0.0, xs[j + 1]), (xs[i + 1], 0.0, xs[j + 1])
            v += [(p00, p01, p11), (p00, p11, p10)]
    # a small box standing on the floor between camera and light casts a shadow along the floor
    c = [(x, y, z) for x in (1.0, 2.0) for y in (0.0, 1.5) for z in (-0.5, 0.5)]
    for q in [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]:
        v += [(c[q[0]], c[q[1]], c[q[2]]), (c[q[0]], c[q[2]], c[q[3]])]
    b.triangles(np.array(v, dtype=np.float64), grey)
    b.background = (0.1, 0.2, 0.3)
    return b.build(), lambda w, h: rtx.camera((-3.0, 6.0, 7.0), (1.0, 0.0, 0.0), (0, 1, 0), 30.0, w, h)


def test_retry_list_overflow_rerenders_with_exact_shadows(rtx, orc):
    """More degenerate feelers in one shadow launch than its retry list holds (2^20 per light): no feeler is dropped and the
    call does not fail; the library notices the overflow flag after the frame's last shadow launch and renders the frame
    again with every shadow launch on the all-f64 restatement of the reference's walk.  The frame must match the
    oracle; a frame with few degenerate feelers (same scene, small) takes the ordinary path and matches too."""
    flat, cam_fn = _floor_scene(rtx)
    osc = orc.Scene(flat)
    sc = rtx.Scene(flat)
    assert sc.info().all_opaque == 1
    for size, ps in (((1280, 1024), 2), ((96, 64), 1)):   # 5.2 M primary samples, most of them on the floor / 6 k
        cam = cam_fn(*size)
        opt = rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=1, pixel_samples=ps, rng_mode=1, seed=4)
        rgba, ids, st = sc.render(cam, opt)
        rows = (size[1] // 4, 3 * size[1] // 4) if size[1] > 64 else (0, size[1])
        orgba, oids, _ = osc.render(cam, opt, threads=16, rows=rows)
        r = slice(*rows)
        assert (ids[r] >= 0).mean() > 0.5
        _compare(rgba[r], ids[r], orgba[r], oids[r])
        assert st.rays_shadow > 0.5 * st.rays_primary
        assert st.exact_shadow_rerenders == (1 if ps == 2 else 0)
    sc.close()


def test_mid_frame_failure_drains_and_handle_stays_usable(rtx):
    """A pool that runs out on a LATER wavefront step, in a scene with three lights (both of the library's shadow streams are busy
    when the step fails): the call returns RTX_ERR_QUEUE_OVERFLOW only after its launches have drained, and the very next
    render on the same handle, into the same output array, is the correct frame."""
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=True, third_light=True)
    assert flat.n_lights == 3
    cam = cam_fn(160, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=8, refract_depth=8, shadow_samples=2, pixel_samples=2, rng_mode=1, seed=3)
    good, ids, st = sc.render(cam, rtx.options(**kw))
    out = np.zeros_like(good)
    with pytest.raises(rtx.abi.RtxError) as err:   # the first steps fit; the pool runs out once the segment stack has grown
        sc.render(cam, rtx.options(batch_rays=16384, pool_rays=16384, **kw), out=out)
    assert err.value.status == rtx.abi.RTX_ERR_QUEUE_OVERFLOW
    again, ids2, st2 = sc.render(cam, rtx.options(**kw), out=out)
    sc.close()
    assert np.array_equal(ids, ids2) and st.rays_total == st2.rays_total
    assert (np.abs(good.astype(int) - again.astype(int)) <= 1).all() and (good != again).mean() < 1e-3


@pytest.mark.parametrize("transmissive,tries", [(False, 16), (False, 4), (False, 1), (True, 2)])
def test_repeated_renders_agree(rtx, transmissive, tries):
    """The same frame rendered five times under the default overlap (three streams, two shadow queues whose group counts, left-over
    lists and retry lists are rewritten every step): a queue reused before its readers are done, or a count read too early, shows as
    a frame that differs from the others.  The f64 accumulators are added to in a different order each time: at most one 8-bit step
    on a handful of pixels; ray counts are equal."""
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=transmissive, third_light=True)
    cam = cam_fn(320, 192)
    sc = rtx.Scene(flat)
    opt = rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=3, batch_rays=32768)
    first, rays = None, None
    for _ in range(5):
        rgba, ids, st = sc.render(cam, opt)
        if first is None:
            first, rays = rgba.copy(), st.rays_total
            assert st.launches_closest > 8    # many steps: the queues alternate
            continue
        d = np.abs(rgba.astype(np.int32) - first.astype(np.int32))
        assert st.rays_total == rays
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (int(d.max()), float((d > 0).mean()))
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tries", [16, 32, 64, 48])
def test_soft_shadows_on_a_ground_plane_that_runs_to_the_horizon(rtx, orc, tries):
    """The bunny stand-in rests on a plane that fills the frame down to the horizon: the hit points on it lie far OUTSIDE the scene's
    cube, and the 16 feelers of such a point towards a sphere light enter the cube far apart.  The cooperative shadow kernel hands
    these groups to the per-lane kernel (list sq.wide; 16, 32 and 64 tries: powers of two; 48: no hand-over, the all-embracing
    beam); the frame must match the oracle either way, and the hand-over must not cost more than the per-lane kernel alone would
    (it used to take 25 times as long: 10.7 ms against 0.4 ms for this frame at full detail)."""
    named = rtx.host.NamedScene("bunny", detail=0.35)
    cam = named.camera(160, 96, 45.0)
    opt = rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=4)
    sc = rtx.Scene(named)
    rgba, ids, st = sc.render(cam, opt)
    rgba, ids, st = sc.render(cam, opt)                                    # (second render: warm)
    _, _, st_lane = sc.render(cam, rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=4,
                                               flags=rtx.abi.FLAG_NO_COOP))
    _, _, st_lane = sc.render(cam, rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=4,
                                               flags=rtx.abi.FLAG_NO_COOP))
    sc.close()
    orgba, oids, ocn = orc.Scene(named).render(cam, opt, threads=16)
    assert (ids == oids).mean() >= 0.9999 and st.rays_total == ocn.rays_total
    assert (np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999
    assert (oids == int(oids.max())).mean() > 0.2                          # the plane (last prim id) fills a good part of the frame
    if tries != 48:
        assert st.ms_shadow <= 3.0 * st_lane.ms_shadow + 0.5, (st.ms_shadow, st_lane.ms_shadow)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["mesh", "mesh_transmissive", "kitchen"])
@pytest.mark.parametrize("tries", [1, 4, 16, 48])
def test_deferred_accumulation_equals_direct(rtx, orc, which, tries, monkeypatch):
    """Where several soft lights share a ray's sample and the shadow launches are long (two sphere lights, 16 tries and more) the shadow
    kernels store each group's result and k_accumulate adds  sum over the lights of weight x shadow  to the sample, one set of atomics
    per ray (DShadowQ.res); elsewhere every launch adds its own.  RTX_DEFER_ACC / RTX_DIRECT_ACC force either: same rays, same
    products, the images differ at most by the order of f64 additions; the deferred frame is also held against the oracle.
    Small batches: many steps, both queues in turn.  48 tries: that launch adds for itself (a group spans waves), the others defer."""
    if which == "kitchen":
        flat, cam_fn = scenes.kitchen_sink_scene(rtx)
    else:
        flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=(which == "mesh_transmissive"))
    cam = cam_fn(160, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=4, refract_depth=4, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=11, batch_rays=8192)
    monkeypatch.setenv("RTX_DEFER_ACC", "1")
    a, ia, sa = sc.render(cam, rtx.options(**kw))
    a2, ia2, sa2 = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_SERIAL, **kw))
    monkeypatch.delenv("RTX_DEFER_ACC")
    monkeypatch.setenv("RTX_DIRECT_ACC", "1")
    b, ib, sb = sc.render(cam, rtx.options(**kw))
    monkeypatch.delenv("RTX_DIRECT_ACC")
    sc.close()
    o, io, so = orc.Scene(flat).render(cam, rtx.options(**kw), threads=16)
    for x, ix, sx in ((a, ia, sa), (a2, ia2, sa2), (b, ib, sb)):
        assert np.array_equal(ix, io) and sx.rays_total == so.rays_total
        assert (np.abs(x[..., :3].astype(int) - o[..., :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3
    assert (np.abs(a.astype(int) - a2.astype(int)) <= 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["mesh", "mesh_transmissive", "kitchen"])
@pytest.mark.parametrize("tries", [1, 3, 4, 16])
def test_sorted_shadow_launches_equal_queue_order(rtx, orc, which, tries, monkeypatch):
    """The per-lane shadow launches of later generations take their groups in the order of the hit points' cells (DShadowQ.cell /
    order, rtx_sort.hip) in scenes with transmissive materials, and the closest-hit launches their rays by direction octant and
    origin cell (DPool.cell, k_closest's `order`: lane i walks ray order[i] and writes that ray's hit record); RTX_SORT_ALL +
    RTX_SORT_OPAQUE force both sorts for every step of every scene (the primaries too, whose cells nobody wrote: any permutation
    must do), RTX_NO_SORT switches them off.  The order of a launch's lanes changes nothing but the order of f64 additions:
    same ids, same ray counts, images within one 8-bit step of each other and of the oracle.  Small batches (many steps, both
    queues in turn; slots past a step's count hold stale cells), overlapped and serial; 3 tries: groups that span waves."""
    if which == "kitchen":
        flat, cam_fn = scenes.kitchen_sink_scene(rtx)
    else:
        flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=(which == "mesh_transmissive"))
    cam = cam_fn(160, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=4, refract_depth=4, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=5, batch_rays=8192)
    no_coop = rtx.abi.FLAG_NO_COOP    # (opaque scenes with 4 or 16 tries: per-lane kernels instead of the cooperative ones)
    monkeypatch.setenv("RTX_SORT_MIN", "0")     # (the default leaves steps of fewer than 65 536 rays alone)
    monkeypatch.setenv("RTX_SORT_ALL", "1")
    monkeypatch.setenv("RTX_SORT_OPAQUE", "1")
    a, ia, sa = sc.render(cam, rtx.options(flags=no_coop, **kw))
    a2, ia2, sa2 = sc.render(cam, rtx.options(flags=no_coop | rtx.abi.FLAG_SERIAL, **kw))
    monkeypatch.delenv("RTX_SORT_ALL")
    monkeypatch.delenv("RTX_SORT_OPAQUE")
    d, id_, sd = sc.render(cam, rtx.options(flags=no_coop, **kw))          # the default: later generations of transmissive scenes
    monkeypatch.setenv("RTX_NO_SORT", "1")
    b, ib, sb = sc.render(cam, rtx.options(flags=no_coop, **kw))
    monkeypatch.delenv("RTX_NO_SORT")
    sc.close()
    o, io, so = orc.Scene(flat).render(cam, rtx.options(**kw), threads=16)
    for x, ix, sx in ((a, ia, sa), (a2, ia2, sa2), (d, id_, sd), (b, ib, sb)):
        assert np.array_equal(ix, io) and sx.rays_total == so.rays_total
        assert (np.abs(x[..., :3].astype(int) - o[..., :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3
    assert (np.abs(a2.astype(int) - b.astype(int)) <= 1).all() and (np.abs(d.astype(int) - b.astype(int)) <= 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["mesh_transmissive", "kitchen"])
def test_computed_primaries_equal_stored_primaries(rtx, orc, which, monkeypatch):
    """A step of primaries stores no ray records: the closest-hit and shading kernels compute the ray of sample s0 + slot where they
    would read the pool (primary_ray; k_closest<MODE, true>, k_shade<GENERAL, true>).  RTX_GENERATE=1 keeps k_generate and the stored
    records.  Same frames either way and against the oracle: several batches of primaries per frame (batch_rays below the sample count:
    s0 > 0), jittered sub-samples, two shards of interleaved tile blocks (samples of foreign blocks are dead), the octree walks too."""
    if which == "kitchen":
        flat, cam_fn = scenes.kitchen_sink_scene(rtx)
    else:
        flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=True)
    cam = cam_fn(168, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=4, pixel_samples=2, rng_mode=1, seed=3, batch_rays=8192)
    o, io, so = orc.Scene(flat).render(cam, rtx.options(**kw), threads=16)
    for flags in (0, rtx.abi.FLAG_OCTREE_WALK, rtx.abi.FLAG_REFERENCE_ORDER):
        a, ia, sa = sc.render(cam, rtx.options(flags=flags, **kw))
        monkeypatch.setenv("RTX_GENERATE", "1")
        b, ib, sb = sc.render(cam, rtx.options(flags=flags, **kw))
        monkeypatch.delenv("RTX_GENERATE")
        assert np.array_equal(ia, io) and np.array_equal(ib, io) and sa.rays_total == so.rays_total == sb.rays_total
        assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3
        assert (np.abs(a[..., :3].astype(int) - o[..., :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999
    # two shards: each renders its own 8-row blocks of the frame
    full = np.zeros_like(o)
    for k in range(2):
        part, _, _ = sc.render(cam, rtx.options(shard_index=k, shard_count=2, **kw))
        for blk in range(k, 96 // 8, 2):
            full[blk * 8:blk * 8 + 8] = part[blk * 8:blk * 8 + 8]
    sc.close()
    assert (np.abs(full[..., :3].astype(int) - o[..., :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999
