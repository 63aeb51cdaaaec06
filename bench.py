#!/usr/bin/env python3
"""bench.py — Mrays/s (all ray types) of the MI355X trace loop on BASELINE.json's metric config.

One "step" = one full frame of the workload (default: the sponza config, 1920x1080, pixel_samples 4
= 16 spp, 16 shadow samples, reflect/refract depth 3, Philox RNG) traced through the C ABI
(rtx_render_device: inputs resident in HBM, outputs to device memory).  With N GPUs the frame's
8-row tile blocks are interleaved over the ranks (block b -> rank b % N), each rank
renders its blocks and the compact block buffers are gathered to rank 0 over RCCL, where they are
de-interleaved into the final Surface (the equivalent of Surface::merge, surface.rs:83-94).  The
total work is fixed as N grows => "scaling": "strong".

Prints ONE JSON line on rank 0.  `roofline` is for the kernel class with the larger share of device
time (shadow feelers on the sponza config).  That kernel works on cache-resident data and is bound
by VALU issue, not by HBM, so the line states the bound it actually runs against:
  bound "valu": frac = VALU-busy x lane utilisation of the kernel running ALONE (rocprofv3 --pmc SQ
  pass, profiles/<tag>_pmc_summary.txt -> profiles/pmc_traffic.json), i.e. the share of the chip's
  16 384 VALU lane-slots per cycle that do a lane's work; next to it hbm_frac (measured HBM bytes per
  launch / launch time alone / 8 TB/s) and l2_frac (128-B L2 requests / 34.5 TB/s).  The average
  launch duration is measured live with HIP events on the launch's own stream.  The figure the
  byte model of SURVEY.md 8d gives (reference-order algorithmic bytes the kernels avoid touching) is
  reported as `work_avoidance_vs_reference_order`, never as a roofline fraction.
`cpu_baseline` is the CPU oracle (literal port of the reference, same thread-pool-of-tiles
structure) timed on a bounded sample of the same frame on this host.

--animate: BASELINE config 5's shape: the frames of a camera animation (default: sibenik, 175 frames =
0-7 s at 25 fps) are dealt over the ranks, frame k on rank k % N, each round of N frames is gathered
to rank 0 (sharding.animate); per-GPU work is fixed as N grows => "scaling": "weak".
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0   # aggregate L2 (8 XCDs x 4 MiB), same guide
VALU_LANE_SLOTS = 256 * 4 * 16   # CUs x SIMDs x lanes issued per cycle

# name -> (scene, width, height, fov, reflect, refract, shadow, gloss, pixel_samples)  SURVEY.md §8d
WORKLOADS = {
    "sphere": ("sphere", 512, 512, 30.0, 4, 4, 1, 1, 1),
    "bunny": ("bunny", 1920, 1080, 45.0, 4, 4, 1, 1, 2),
    "sponza": ("sponza", 1920, 1080, 67.5, 3, 3, 16, 1, 4),
    # BASELINE config 4 names depth 16: the ray tree grows ~x7 per two levels (634 rays per pixel at depth 8, 336 632 at depth 14), so a
    # 1080p frame at depth 16 is ~5e12 rays: hours.  The bench default is depth 8 at 1080p; depth 16 is covered by the parity test
    # (tests/test_gpu_configs.py) and can be benchmarked on a small frame: --reflect 16 --refract 16 --width 64 --height 40
    "heptoroid": ("heptoroid-refractive", 1920, 1080, 45.0, 8, 8, 1, 1, 1),
    "sibenik": ("sibenik", 1920, 1080, 45.0, 3, 3, 4, 1, 2),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sponza", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--detail", type=float, default=1.0, help="tessellation scale of the stand-in meshes")
    ap.add_argument("--batch-rays", type=int, default=0)
    ap.add_argument("--cpu-blocks", type=int, default=3, help="8-row tile blocks the CPU baseline traces (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--reflect", type=int, default=-1, help="override reflect_depth")
    ap.add_argument("--refract", type=int, default=-1, help="override refract_depth")
    ap.add_argument("--no-coop", action="store_true", help="RTX_FLAG_NO_COOP: per-lane octree shadows (A/B)")
    ap.add_argument("--serial", action="store_true", help="RTX_FLAG_SERIAL: no overlap of shadow and closest-hit launches (A/B)")
    ap.add_argument("--animate", action="store_true", help="camera animation, frame k on rank k %% N (BASELINE config 5: --workload sibenik)")
    ap.add_argument("--frames", type=int, default=0, help="--animate: render only the first K frames of the animation")
    ap.add_argument("--fps", type=float, default=25.0)
    ap.add_argument("--time-slice", type=float, nargs=2, default=(0.0, 7.0))
    ap.add_argument("--force", action="store_true", help="run a heptoroid configuration whose estimated ray count is above 1e11")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the trace path has no CPU fallback")
    # Rehearsal switches for a one-GPU box (never set by the driver): RTX_BENCH_BACKEND=gloo runs the N > 1 flow with
    # every rank on cuda:0 and the gather / reductions on host tensors, to exercise everything but RCCL itself.
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    # the process's one-off HIP runtime set-up (context, first allocation and copy: ~190 ms, tools/exp/h2d_cold.cpp) is not scene
    # build: a host application that renders has it behind it.  scene_build_upload_s below still holds the library's own first-use
    # costs (code object load, streams, pinned buffers); scene_build_upload_warm_s is the same call again.
    torch.zeros(1, device=dev).cpu()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("rust-raytracer_amd")
    scene_name, W, H, fov, refl, refr, shadow, gloss, ps = WORKLOADS[args.workload]
    W = args.width or W
    H = args.height or H
    refl = args.reflect if args.reflect >= 0 else refl
    refr = args.refract if args.refract >= 0 else refr
    if args.workload == "heptoroid" and not args.force:
        est = W * H * ps * ps * 634.0 * 7.0 ** ((min(refl, refr) - 8) / 2.0)    # measured: tools/gpu_hept.py, DESIGN.md section 9
        if est > 1e11:
            raise SystemExit("heptoroid-refractive at depth %d/%d and %dx%d is about %.1e rays (x7 per two levels): hours on one GPU. "
                             "Use a smaller frame (--width 64 --height 40), a lower depth, or --force." % (refl, refr, W, H, est))
    if args.animate:
        return animate(args, pkg, torch, dist, np, world, rank, local_rank, dev, comm_dev, scene_name, W, H, fov, refl, refr, shadow, gloss, ps)

    t0 = time.time()
    named = pkg.host.NamedScene(scene_name, detail=args.detail)   # the scene file's work: procedural stand-in mesh (the reference loads an OBJ here)
    cam = named.camera(W, H, fov)
    generate_s = time.time() - t0
    t0 = time.time()
    scene = pkg.Scene(named, device=local_rank)                   # rtx_scene_create: octree, box lists, upload, f32 twins on the device
    info = scene.info()
    build_s = time.time() - t0
    # the same call again: without the library's first-use costs
    t0 = time.time()
    pkg.Scene(named, device=local_rank).close()
    build_warm_s = time.time() - t0

    n_blocks = H // 8
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    gat = pkg.sharding.StillGather(W, H, rank, world, comm_dev)
    final = gat.final

    opt = pkg.options(reflect_depth=refl, refract_depth=refr, shadow_samples=shadow, gloss_samples=gloss,
                      pixel_samples=ps, rng_mode=pkg.abi.RNG_PHILOX, seed=0, frame=0,
                      shard_index=rank, shard_count=world, batch_rays=args.batch_rays,
                      flags=(pkg.abi.FLAG_NO_COOP if args.no_coop else 0) | (pkg.abi.FLAG_SERIAL if args.serial else 0))
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        st = scene.render_device(cam, opt, rgba.data_ptr(), None, stream, want_stats=True)
        gat.gather(rgba if comm_dev is dev else rgba.cpu(), dist)   # RCCL over xGMI: <= 1.04 MB per rank at 1080p (a copy at N=1)
        return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t_start

    rays = np.zeros(5, dtype=np.float64)
    for st in stats:
        rays += np.array([st.rays_primary, st.rays_reflect, st.rays_gloss, st.rays_refract, st.rays_shadow], dtype=np.float64)
    agg = torch.tensor(list(rays) + [elapsed], dtype=torch.float64, device=comm_dev)
    if world > 1:
        tmax = agg[5:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg[:5], op=dist.ReduceOp.SUM)
        agg[5] = tmax[0]
    agg = agg.cpu().numpy()
    total_rays, elapsed = float(agg[:5].sum()), float(agg[5])

    if rank == 0:
        steps = max(1, args.steps)
        ms_closest = sum(s.ms_closest for s in stats) / steps
        ms_shadow = sum(s.ms_shadow for s in stats) / steps
        l_closest = sum(s.launches_closest for s in stats) / steps
        l_shadow = sum(s.launches_shadow for s in stats) / steps
        r0 = stats[-1]
        closest_rays = r0.rays_primary + r0.rays_reflect + r0.rays_gloss + r0.rays_refract
        out = {
            "metric": "Mrays/sec (all ray types)", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %dx%d pixel_samples %d (%d spp) shadow_samples %d reflect %d refract %d, "
                                   "procedural stand-in mesh (reference assets absent), philox seed 0" %
                                   (scene_name, W, H, ps, ps * ps, shadow, refl, refr),
                       "triangles": int(named.n_triangles), "octree_depth": int(info.octree_depth),
                       "octree_nodes": int(info.n_nodes), "leaf_entries": int(info.n_leaf_entries),
                       "sharding": "8-row tile blocks interleaved over ranks + RCCL gather to rank 0",
                       "scene_generate_s": round(generate_s, 3), "scene_build_upload_s": round(build_s, 3),
                       "scene_build_upload_warm_s": round(build_warm_s, 3)},
            "frame_ms": elapsed / steps * 1e3,
            "rays_per_frame": {"primary": float(agg[0]) / steps, "reflect": float(agg[1]) / steps,
                               "gloss": float(agg[2]) / steps, "refract": float(agg[3]) / steps,
                               "shadow": float(agg[4]) / steps},
            "device_ms_rank0": {"frame": sum(s.ms_device for s in stats) / steps, "closest_shade": ms_closest,
                                "shadow": ms_shadow, "launches_closest": l_closest, "launches_shadow": l_shadow},
        }

        wk = [[int(r0.work[c][k]) for k in range(3)] for c in range(2)]
        if any(any(w) for w in wk):   # RTX_LIB=librtx_count.so: what the kernels actually walked, per ray
            out["gpu_work_per_ray"] = {
                "closest": dict(zip(("nodes", "leaf_entries", "prim_tests"), [w / max(1, closest_rays) for w in wk[0]])),
                "shadow": dict(zip(("nodes", "leaf_entries", "prim_tests"), [w / max(1, r0.rays_shadow) for w in wk[1]]))}

        cpu = None
        if args.cpu_blocks > 0 and world == 1:   # the CPU baseline (and the roofline built on its counters) is an N=1 figure
            orc = importlib.import_module("oracle.oracle")   # the checker + CPU baseline, never the product
            threads = args.cpu_threads or int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            oscene = orc.Scene(named)
            ocn = orc.Counters()
            opt_cpu = pkg.options(reflect_depth=refl, refract_depth=refr, shadow_samples=shadow, gloss_samples=gloss,
                                  pixel_samples=ps, rng_mode=pkg.abi.RNG_PHILOX, seed=0, frame=0)
            picks = sorted({int((k + 0.5) * n_blocks / args.cpu_blocks) for k in range(args.cpu_blocks)})
            gpu_img = final.cpu().numpy()
            ok = tot = 0
            for b in picks:
                o_rgba, _, cn = oscene.render(cam, opt_cpu, threads=threads, rows=(b * 8, b * 8 + 8), cols=(0, W), want_ids=False)
                ocn.add(cn)
                d = np.abs(o_rgba[b * 8:b * 8 + 8, :, :3].astype(np.int32) - gpu_img[b * 8:b * 8 + 8, :, :3].astype(np.int32)).max(axis=-1)
                ok += int((d <= 1).sum())
                tot += d.size
            cpu = {"value": ocn.rays_total / ocn.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",   # a SAMPLE of the frame:
                   "sample": "tile-block rows %s (8 rows x %d px each) of the same %dx%d frame, same options; %d rays in %.1f s"
                             % (picks, W, W, H, ocn.rays_total, ocn.seconds)}
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = out["value"] / cpu["value"]
            out["parity_sample"] = {"rgb_within_1": ok / max(1, tot), "pixels": tot}
            # reference-order algorithmic bytes per ray of each class
            b_closest = ocn.algorithmic_bytes(0) / max(1, ocn.rays_closest)
            b_shadow = ocn.algorithmic_bytes(1) / max(1, ocn.rays[4])
            if ms_shadow >= ms_closest and l_shadow > 0:
                kname, ms, launches, nrays, bpr = "k_shadow", ms_shadow, l_shadow, float(r0.rays_shadow), b_shadow
            else:
                kname, ms, launches, nrays, bpr = "k_trace", ms_closest, l_closest, float(closest_rays), b_closest
            avg_ms = ms / max(1.0, launches)
            # what the byte model of SURVEY.md 8d gives: reference-order algorithmic bytes (oracle traversal counters: every pierced
            # leaf, no early-out) x rays / launch time / HBM peak.  The kernels never move those bytes: a work-avoidance figure.
            ref_order_gbs = bpr * (nrays / max(1.0, launches)) / (avg_ms * 1e-3) / 1e9
            traffic = None
            pmc = {}
            try:   # HBM bytes per launch of that kernel + SQ counters from separate rocprofv3 --pmc passes (profiles/)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    allpmc = json.load(fh)
                    pmc = allpmc.get(args.workload, {})
                    traffic = pmc.get(kname)
                    pmc_source = allpmc.get("_source")
            except (OSError, ValueError):
                pmc_source = None
            v = pmc.get(kname + "_valu") or {}
            frac = (v.get("busy") * v.get("lane_utilisation")) if v else None
            out["roofline"] = {
                "bound": "valu", "kernel": {"k_shadow": "k_shadow_coop<false>", "k_trace": "k_closest<5> (+ k_shade)"}[kname],
                "achieved": frac * VALU_LANE_SLOTS if frac is not None else None, "peak": VALU_LANE_SLOTS,
                "unit": "VALU lane-slots per cycle doing a lane's work (256 CUs x 4 SIMDs x 16 lanes)",
                "frac": frac,
                "valu_busy": v.get("busy"), "lane_utilisation": v.get("lane_utilisation"),
                "valu_insts_per_wave": v.get("valu_insts_per_wave"),
                "hbm_frac": (v.get("hbm_gbs_alone") / HBM_PEAK_GBS) if v else None,
                "l2_frac": (v.get("l2_request_gbs") / L2_PEAK_GBS) if v else None,
                "hbm_gbs_alone": v.get("hbm_gbs_alone"), "l2_request_gbs": v.get("l2_request_gbs"), "l2_hit_rate": v.get("l2_hit_rate"),
                "traffic": traffic,   # measured HBM bytes per launch: (FETCH_SIZE x2 + WRITE_SIZE) x 1024 / launches
                "avg_launch_ms": avg_ms,   # live, HIP events on the launch's own stream
                "avg_launch_ms_alone_profiled": v.get("alone_ms_per_launch"),
                "launches_per_frame": launches, "rays_per_launch": nrays / max(1.0, launches),
                # > 1: launches run concurrently (the shadow launches of the two lights on the library's two
                # streams, next to the following step's closest-hit launch), so avg_launch_ms is a duration
                # under sharing; `bench.py --serial` times every kernel alone (profiles/*_serial*)
                "launch_concurrency": (ms_shadow + ms_closest) / max(1e-9, out["device_ms_rank0"]["frame"]),
                "overlap": not args.serial,
                "algorithmic_bytes_per_ray_reference_order": bpr,
                "work_avoidance_vs_reference_order": ref_order_gbs / HBM_PEAK_GBS,
                "source": pmc_source,
                "note": "the dominant kernel is VALU-issue bound on cache-resident data: frac = valu_busy x lane_utilisation from the "
                        "rocprofv3 SQ pass of the kernel alone (committed under profiles/, not collected in this run); hbm_frac and "
                        "l2_frac say how far it is from the memory bounds; work_avoidance_vs_reference_order is the SURVEY.md 8d byte "
                        "model (bytes the reference's unordered walk would move per ray x rays / launch time / 8 TB/s), > 1 because the "
                        "ordered, culled, cooperative traversal does not move them: not a roofline fraction"}
        print(json.dumps(out), flush=True)

    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def animate(args, pkg, torch, dist, np, world, rank, local_rank, dev, comm_dev, scene_name, W, H, fov, refl, refr, shadow, gloss, ps):
    """BASELINE config 5: frames of the camera animation dealt over the ranks (frame k -> rank k % N), one gather per round of
    N frames to rank 0 (Animator::animate, animator.rs:21-56; sharding.animate).  A step = one round."""
    import math
    sh = pkg.sharding
    t0 = time.time()
    named = pkg.host.NamedScene(scene_name, detail=args.detail)
    scene = pkg.Scene(named, device=local_rank)
    build_s = time.time() - t0
    frames = sh.animation_frames(args.fps, args.time_slice[0], args.time_slice[1])
    if args.frames > 0:
        frames = frames[:args.frames]
    if len(frames) < world:
        raise SystemExit("--animate needs at least one frame per rank (%d frames, %d ranks)" % (len(frames), world))
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rays = np.zeros(5, dtype=np.float64)
    ms_dev, ms_call = [], []
    last = {}

    def render_frame(k, t):
        cam = named.camera(W, H, fov, time=t)      # Animator::lerp_camera (animator.rs:91-114) in the C++ mirror
        opt = pkg.options(reflect_depth=refl, refract_depth=refr, shadow_samples=shadow, gloss_samples=gloss, pixel_samples=ps,
                          rng_mode=pkg.abi.RNG_PHILOX, seed=0, frame=k, batch_rays=args.batch_rays,
                          flags=(pkg.abi.FLAG_NO_COOP if args.no_coop else 0) | (pkg.abi.FLAG_SERIAL if args.serial else 0))
        st = scene.render_device(cam, opt, rgba.data_ptr(), None, stream, want_stats=True)
        rays[:] += [st.rays_primary, st.rays_reflect, st.rays_gloss, st.rays_refract, st.rays_shadow]
        ms_dev.append(st.ms_device)
        ms_call.append(st.ms_total)
        return rgba if comm_dev is dev else rgba.cpu()

    def on_frame(k, t, frame):      # rank 0: where the reference's writer thread would write {output_file}{:06}.ppm
        last["k"] = k
        last["sum"] = int(frame[::37, ::41].sum().item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):   # with on_frame: its first call loads torch's reduction kernel (~70 ms), not part of a frame
        sh.animate(frames[:world], rank, world, dist, render_frame, on_frame)
    rays[:] = 0
    del ms_dev[:]
    del ms_call[:]
    barrier()
    t_start = time.perf_counter()
    n_mine = sh.animate(frames, rank, world, dist, render_frame, on_frame)
    barrier()
    elapsed = time.perf_counter() - t_start
    agg = torch.tensor(list(rays) + [elapsed, float(n_mine)], dtype=torch.float64, device=comm_dev)
    if world > 1:
        tmax = agg[5:6].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        summed = agg.clone()
        dist.all_reduce(summed, op=dist.ReduceOp.SUM)
        agg = summed
        agg[5] = tmax[0]
    agg = agg.cpu().numpy()
    total_rays, elapsed = float(agg[:5].sum()), float(agg[5])
    rounds = math.ceil(len(frames) / world)
    if rank == 0:
        info = scene.info()
        out = {"metric": "Mrays/sec (all ray types)", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
               "steps": rounds, "warmup": args.warmup, "ms_per_step": elapsed / rounds * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s camera animation %g-%g s at %g fps: %d frames of %dx%d, pixel_samples %d (%d spp), shadow_samples %d, "
                                      "reflect %d refract %d, frame k on rank k %% %d, one gather per round of %d frames to rank 0; procedural "
                                      "stand-in mesh (reference assets absent), philox seed 0" %
                                      (scene_name, args.time_slice[0], args.time_slice[1], args.fps, len(frames), W, H, ps, ps * ps, shadow,
                                       refl, refr, world, world),
                          "triangles": int(named.n_triangles), "octree_nodes": int(info.n_nodes), "scene_build_upload_s": round(build_s, 3)},
               "frames": len(frames), "frames_rank0": int(n_mine), "ms_per_frame_node": elapsed / len(frames) * 1e3,
               "ms_per_frame_rank0_device": sum(ms_dev) / max(1, len(ms_dev)),
               "ms_per_frame_rank0_call": sum(ms_call) / max(1, len(ms_call)), "frames_per_s_node": len(frames) / elapsed,
               "rays_per_frame": total_rays / len(frames), "last_frame_gathered": last.get("k")}
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
