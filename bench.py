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
    ap.add_argument("--pmc", default="auto", choices=("auto", "profile", "off"),
                    help="roofline counters: auto = collect them now (rocprofv3 --pmc passes on child runs of this workload; falls back to the "
                         "committed profile), profile = profiles/pmc_traffic.json only, off = no counters")
    ap.add_argument("--save-profile", default="", help="TAG: keep the counter passes of this run as <profile-dir>/TAG_pmc_<workload>.json and in <profile-dir>/pmc_traffic.json")
    ap.add_argument("--profile-dir", default=os.path.join(ROOT, "profiles"))
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # a child run under the profiler / the counting library: no extras
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
    # every rank renders ONLY its own 8-row blocks, packed (RTX_FLAG_COMPACT_OUTPUT), straight into the buffer the gather sends:
    # H/N rows per rank instead of a full frame, no pack step
    gat = pkg.sharding.StillGather(W, H, rank, world, comm_dev)
    final = gat.final
    target = gat.target if comm_dev is dev else torch.zeros_like(gat.target, device=dev)
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)    # full-frame buffer of measure()'s single-GPU extra frames

    base_flags = (pkg.abi.FLAG_NO_COOP if args.no_coop else 0) | (pkg.abi.FLAG_SERIAL if args.serial else 0)
    opt = pkg.options(reflect_depth=refl, refract_depth=refr, shadow_samples=shadow, gloss_samples=gloss,
                      pixel_samples=ps, rng_mode=pkg.abi.RNG_PHILOX, seed=0, frame=0,
                      shard_index=rank, shard_count=world, batch_rays=args.batch_rays,
                      flags=base_flags | pkg.abi.FLAG_COMPACT_OUTPUT)
    stream = torch.cuda.current_stream().cuda_stream

    def step(want_stats=False):
        st = scene.render_device(cam, opt, target.data_ptr(), None, stream, want_stats=want_stats)
        if comm_dev is not dev:
            gat.target.copy_(target)                                # gloo rehearsal: the collective runs on host tensors
        gat.gather(dist)                                            # RCCL over xGMI: <= 1.04 MB per rank at 1080p (a de-interleave at N=1)
        return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    # The timed frames run as a caller's would, without rtx_stats (its per-launch timing events cost 0.05 ms of the 0.21 ms sphere frame,
    # 2 % of bunny's).  Ray counts and per-kernel device times come from the SAME frame rendered again, untimed: the counts of a frame
    # are a function of the scene, camera, options and seed only.
    n_stat = max(1, min(2, args.steps))
    stats = [step(want_stats=True) for _ in range(n_stat)]
    barrier()

    rays = np.zeros(5, dtype=np.float64)
    for st in stats:
        rays += np.array([st.rays_primary, st.rays_reflect, st.rays_gloss, st.rays_refract, st.rays_shadow], dtype=np.float64)
    assert all(st.rays_total == stats[0].rays_total for st in stats)
    rays *= args.steps / float(n_stat)
    agg = torch.tensor(list(rays) + [elapsed], dtype=torch.float64, device=comm_dev)
    if world > 1:
        tmax = agg[5:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg[:5], op=dist.ReduceOp.SUM)
        agg[5] = tmax[0]
    agg = agg.cpu().numpy()
    # each rank's own frame time (render + its share of the gather): load balance of the interleaved blocks
    mine_ms = torch.tensor([elapsed / max(1, args.steps) * 1e3], dtype=torch.float64, device=comm_dev)
    if world > 1:
        all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)
        per_rank = [float(t.item()) for t in all_ms]
    else:
        per_rank = [float(mine_ms.item())]
    per_rank_ms = {"min": min(per_rank), "max": max(per_rank), "by_rank": [round(x, 3) for x in per_rank]}
    total_rays, elapsed = float(agg[:5].sum()), float(agg[5])

    if rank == 0:
        steps = max(1, args.steps)
        ms_closest = sum(s.ms_closest for s in stats) / n_stat
        ms_shade = sum(s.ms_shade for s in stats) / n_stat
        ms_shadow = sum(s.ms_shadow for s in stats) / n_stat
        l_closest = sum(s.launches_closest for s in stats) / n_stat
        l_shadow = sum(s.launches_shadow for s in stats) / n_stat
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
                       "hierarchy_built_on_device": bool(info.hierarchy_on_device),
                       "scene_generate_s": round(generate_s, 3), "scene_build_upload_s": round(build_s, 3),
                       "scene_build_upload_warm_s": round(build_warm_s, 3)},
            "frame_ms": elapsed / steps * 1e3,
            "rays_per_frame": {"primary": float(agg[0]) / steps, "reflect": float(agg[1]) / steps,
                               "gloss": float(agg[2]) / steps, "refract": float(agg[3]) / steps,
                               "shadow": float(agg[4]) / steps},
            "stats_from": "%d untimed frame(s) rendered after the timed region with rtx_stats (the timed frames carry no instrumentation)" % n_stat,
            "device_ms_rank0": {"frame": sum(s.ms_device for s in stats) / n_stat, "closest_shade": ms_closest, "shade": ms_shade,
                                "shadow": ms_shadow, "launches_closest": l_closest, "launches_shadow": l_shadow},
            "per_rank_frame_ms": per_rank_ms,
        }

        wk = [[int(r0.work[c][k]) for k in range(3)] for c in range(2)]
        if any(any(w) for w in wk):   # RTX_LIB=librtx_count.so: what the kernels actually walked, per ray
            out["gpu_work_per_ray"] = {
                "closest": dict(zip(("nodes", "leaf_entries", "prim_tests"), [w / max(1, closest_rays) for w in wk[0]])),
                "shadow": dict(zip(("nodes", "leaf_entries", "prim_tests"), [w / max(1, r0.rays_shadow) for w in wk[1]]))}

        if world == 1 and not args.child:
            measure(args, pkg, np, torch, out, scene, named, cam, opt, rgba, final, stream, W, H, n_blocks,
                    refl, refr, shadow, gloss, ps, stats, closest_rays)
        print(json.dumps(out), flush=True)

    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64, /opt/skills/guides/MI355X_MICROARCH.md
PMC_PASSES = (   # separate rocprofv3 --pmc runs, each within one pass's counter limits (8 SQ, 4 TCC, 2 GRBM)
    ("f", ["FETCH_SIZE", "TCC_HIT_sum"]),
    ("w", ["WRITE_SIZE", "TCC_MISS_sum", "TCC_REQ_sum"]),
    ("s", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_WAIT_ANY",
           "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]),
)


def reference_num_cpus():
    """What num_cpus::get() gives the reference (main.rs:138 sizes its thread pool with it): the logical CPUs this process may use =
    min(scheduler affinity, cgroup CPU quota)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / pr)))
        except (OSError, ValueError):
            pass
    return n


def short_kernel_name(name):
    """'void k_shadow_coop<false>(DScene, ...)' -> 'k_shadow_coop<false>' (the names rtx_last_kernel_names reports)."""
    import re
    nm = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void\s+)?([\w:]+(?:<[^()]*?>)?)\s*\(", nm)
    short = m.group(1) if m else nm.split("(")[0].strip()
    # k_closest<MODE, PRIM> / k_shade<GENERAL, PRIM>: the instance for a step of primaries (which computes its rays) and the one for pool
    # rays are one kernel to the accounting, as they are to rtx_last_kernel_names
    m2 = re.match(r"(k_closest|k_shade)<([^,>]+),\s*(?:true|false)>$", short)
    return "%s<%s>" % (m2.group(1), m2.group(2)) if m2 else short


def child_cmd(args, extra):
    cmd = ["python3", os.path.join(ROOT, "bench.py"), "--child", "--pmc", "off", "--cpu-blocks", "0", "--workload", args.workload,
           "--detail", str(args.detail), "--batch-rays", str(args.batch_rays), "--reflect", str(args.reflect), "--refract", str(args.refract)]
    if args.width:
        cmd += ["--width", str(args.width)]
    if args.height:
        cmd += ["--height", str(args.height)]
    if args.no_coop:
        cmd.append("--no-coop")
    if args.force:
        cmd.append("--force")
    return cmd + extra


def collect_pmc(args):
    """Three rocprofv3 --pmc passes over `bench.py --serial --steps 1 --warmup 0` of this workload (child processes; the profiler
    serialises dispatches, so every figure is the kernel running alone).  Returns {kernel: {counter: sum, '_seconds', '_dispatches'}}
    per pass, or raises."""
    import collections
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="rtx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RTX_LIB", None)
    res = {}
    for tag, counters in PMC_PASSES:
        cmd = ["timeout", "-s", "KILL", "150", rocprof, "--pmc"] + counters + ["-d", os.path.join(tmp, tag), "-o", tag, "--"] + \
              child_cmd(args, ["--steps", "1", "--warmup", "1", "--serial"])   # (the kernels of the frame that is timed alone; the warm-up
        # frame is a handle's probe frame, which may run another instance of the cooperative shadow kernel: rtx_scene::coop_plain_frames)
        pr = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if pr.returncode != 0:
            raise RuntimeError("rocprofv3 pass %s failed (%d): %s" % (tag, pr.returncode, pr.stdout[-400:]))
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(tmp, tag)) for f in fs if f.endswith(".db")]
        if not dbs:
            raise RuntimeError("rocprofv3 pass %s left no database" % tag)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        con = sqlite3.connect(dbs[0])
        seen = set()
        for name, did, cname, val, start, end in con.execute(
                "select kernel_name, dispatch_id, counter_name, value, start, end from counters_collection"):
            k = short_kernel_name(name)
            agg[k][cname] += float(val)
            if did not in seen:
                seen.add(did)
                agg[k]["_seconds"] += (int(end) - int(start)) * 1e-9
                agg[k]["_dispatches"] += 1
        con.close()
        res[tag] = {k: dict(v) for k, v in agg.items()}
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def derive_pmc(res):
    """Per kernel, from the three passes (units and the gfx950 corrections as the micro-architecture guide prescribes: FETCH_SIZE /
    WRITE_SIZE in KB, FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2; SQ_ACTIVE_INST_* in quad-cycles; GRBM_GUI_ACTIVE summed
    over the 8 XCDs)."""
    out = {}
    f, w, s = res["f"], res["w"], res["s"]
    total_s = sum(c.get("_seconds", 0.0) for c in s.values()) or 1.0
    for k, c in s.items():
        if k not in f or k not in w or not c.get("SQ_WAVES"):
            continue
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
        launches = max(1.0, c["_dispatches"])
        hbm_bytes = (f[k].get("FETCH_SIZE", 0.0) * 2 + w[k].get("WRITE_SIZE", 0.0)) * 1024
        busy = c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles if simd_cycles else 0.0
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) if c.get("SQ_ACTIVE_INST_VALU") else 0.0
        out[k] = {"busy": busy, "lane_utilisation": lanes,
                  "valu_insts_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
                  # the scalar unit is ONE per CU (shared by its four SIMDs): instructions issued per CU-cycle, peak 1
                  "salu_insts_per_wave": c.get("SQ_INSTS_SALU", 0.0) / c["SQ_WAVES"],
                  "salu_issue_per_cu_cycle": c.get("SQ_INSTS_SALU", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8 * 256) if c.get("GRBM_GUI_ACTIVE") else None,
                  "waves_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                  "l2_hit_rate": 1.0 - w[k]["TCC_MISS_sum"] / w[k]["TCC_REQ_sum"] if w[k].get("TCC_REQ_sum") else None,
                  "l2_request_gbs": w[k].get("TCC_REQ_sum", 0.0) * 128 / w[k]["_seconds"] / 1e9,
                  "hbm_gbs_alone": hbm_bytes / w[k]["_seconds"] / 1e9,
                  "hbm_bytes_per_launch": hbm_bytes / launches,
                  "hbm_write_bytes_per_launch": w[k].get("WRITE_SIZE", 0.0) * 1024 / launches,
                  "alone_ms_per_launch": c["_seconds"] * 1e3 / launches, "launches": launches,
                  "share_of_kernel_time": c["_seconds"] / total_s}
    return out


def count_work(args):
    """What the kernels themselves walked: one frame through librtx_count.so (the profiling twin, -DRTX_COUNT_WORK) in a child
    process; returns its gpu_work_per_ray."""
    import subprocess
    env = dict(os.environ, RTX_LIB="librtx_count.so")
    pr = subprocess.run(child_cmd(args, ["--steps", "1", "--warmup", "0"]), cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        text=True, timeout=280)
    if pr.returncode != 0:
        raise RuntimeError("count run failed: " + pr.stderr[-300:])
    return json.loads(pr.stdout.strip().splitlines()[-1]).get("gpu_work_per_ray")


def useful_flops_per_ray(kernel, work):
    """SURVEY.md 8d's flop weights (12 per box test, 55 per primitive test) x the kernel's OWN counters (librtx_count.so):
    k_shadow_coop: a lane tests one child box per node its group visits (`nodes`), then the queued prims' boxes against its own ray
    (`leaf_entries`); per-lane walks over the 4-wide hierarchy test 4 child boxes per node visit; the octree walks 8 child boxes per
    node and one box per leaf entry."""
    if not work:
        return None
    cls = "closest" if kernel.startswith("k_closest") else ("shadow" if kernel.startswith("k_shadow") else None)
    if cls is None:
        return None
    w = work[cls]
    if kernel.startswith("k_shadow_coop"):
        box = w["nodes"] + w["leaf_entries"]
    elif kernel in ("k_closest<5>", "k_closest<8>", "k_shadow<6>", "k_shadow<7>"):
        box = 4.0 * w["nodes"]
    else:
        box = 8.0 * w["nodes"] + w["leaf_entries"]
    return 12.0 * box + 55.0 * w["prim_tests"]


def measure(args, pkg, np, torch, out, scene, named, cam, opt, rgba, final, stream, W, H, n_blocks, refl, refr, shadow, gloss, ps, stats, closest_rays):
    """cpu_baseline + roofline of the N = 1 line (rank 0)."""
    steps = max(1, len(stats))   # (the instrumented frames rendered after the timed region)
    r0 = stats[-1]
    names = scene.last_kernel_names()
    out["kernels"] = names
    # ---- every kernel ALONE, live: one more frame with RTX_FLAG_SERIAL (HIP events on the launches' own stream)
    opt_serial = pkg.options(reflect_depth=refl, refract_depth=refr, shadow_samples=shadow, gloss_samples=gloss, pixel_samples=ps,
                             rng_mode=pkg.abi.RNG_PHILOX, seed=0, frame=0, batch_rays=args.batch_rays,
                             flags=(pkg.abi.FLAG_NO_COOP if args.no_coop else 0) | pkg.abi.FLAG_SERIAL)
    scene.render_device(cam, opt_serial, rgba.data_ptr(), None, stream, want_stats=True)   # (its own warm-up: first use of the serial path)
    sa = scene.render_device(cam, opt_serial, rgba.data_ptr(), None, stream, want_stats=True)
    names_serial = scene.last_kernel_names()
    n_lights = sum(1 for k in names if k.startswith("shadow"))
    lc = max(1, sa.launches_closest)
    alone = {}   # kernel -> [ms per frame alone, launches per frame, rays per frame]
    alone[names_serial["closest"]] = [sa.ms_closest - sa.ms_shade, lc, float(closest_rays)]
    alone[names_serial["shade"]] = [sa.ms_shade, lc, float(closest_rays)]
    for li in range(n_lights):
        k = names_serial.get("shadow%d" % li, "")
        if k:
            e = alone.setdefault(k, [0.0, 0, 0.0])
            e[0] += sa.ms_shadow_light[li]
            e[1] += lc
    shadow_kernels = [k for k in alone if k.startswith("k_shadow")]
    for k in shadow_kernels:   # feelers per kernel: the launches' share of the frame's shadow launches (exact with one shadow kernel)
        alone[k][2] = float(r0.rays_shadow) * alone[k][1] / max(1, sum(alone[q][1] for q in shadow_kernels))
    dominant = max(alone, key=lambda k: alone[k][0])
    out["kernel_ms_alone_live"] = {k: {"ms_per_frame": v[0], "launches_per_frame": v[1], "ms_per_launch": v[0] / max(1, v[1])} for k, v in alone.items()}

    # ---- CPU baseline: the oracle (literal port of the reference, same 128x8 tile pool) on a bounded sample of the same frame
    cpu = None
    if args.cpu_blocks > 0:
        orc = importlib.import_module("oracle.oracle")   # the checker + CPU baseline, never the product
        ref_cpus = reference_num_cpus()
        share = int(os.environ.get("OMP_NUM_THREADS", "0")) or ref_cpus   # the GPU box sets its per-job CPU share here (16)
        threads = args.cpu_threads or max(1, min(ref_cpus, share))
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
        cpu = {"value": ocn.rays_total / ocn.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "host_cores": os.cpu_count(),
               "reference_num_cpus": ref_cpus, "kind": "port",
               "threads_note": "the reference sizes its pool with num_cpus::get() (main.rs:138) = reference_num_cpus; cores = min(that, the job's CPU "
                               "share OMP_NUM_THREADS=%s)" % os.environ.get("OMP_NUM_THREADS", "unset"),
               "sample": "tile-block rows %s (8 rows x %d px each) of the same %dx%d frame, same options; %d rays in %.1f s"
                         % (picks, W, W, H, ocn.rays_total, ocn.seconds)}
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = out["value"] / cpu["value"]
        out["parity_sample"] = {"rgb_within_1": ok / max(1, tot), "pixels": tot}

    # ---- counters of the dominant kernel: collected now, else the committed profile of the same command
    pmc, pmc_source = None, None
    os.makedirs(args.profile_dir, exist_ok=True)
    profile_path = os.path.join(args.profile_dir, "pmc_traffic.json")
    if args.pmc == "auto":
        try:
            raw = collect_pmc(args)
            pmc = derive_pmc(raw)
            pmc_source = "live: rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE TCC_REQ | SQ_* GRBM_GUI_ACTIVE) on child runs of this command"
            if args.save_profile:
                with open(os.path.join(args.profile_dir, "%s_pmc_%s.json" % (args.save_profile, args.workload)), "w") as fh:
                    json.dump({"command": "bench.py --workload %s --steps 1 --warmup 0 (rocprofv3 --pmc, three passes)" % args.workload,
                               "passes": {t: c for t, c in PMC_PASSES}, "raw_sums_per_kernel": raw, "derived": pmc}, fh, indent=1)
                allp = {}
                try:
                    allp = json.load(open(profile_path))
                except (OSError, ValueError):
                    pass
                allp["_source"] = ("profiles/<tag>_pmc_<workload>.json (bench.py --save-profile: separate rocprofv3 --pmc passes): bytes per launch = "
                                   "(FETCH_SIZE x2 + WRITE_SIZE) KB x 1024 / launches; busy = SQ_ACTIVE_INST_VALU x4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)")
                allp[args.workload] = {"_tag": args.save_profile,   # (the trace kernels; the per-workload file keeps every kernel of the run)
                                       "kernels": {k: v for k, v in pmc.items() if k.startswith(("k_closest", "k_shade", "k_shadow", "k_generate", "k_resolve", "k_group"))}}
                json.dump(allp, open(profile_path, "w"), indent=1)
        except Exception as e:   # noqa: BLE001  (no profiler, a refused pass, a timeout: the committed profile)
            out["pmc_live_error"] = str(e)[:300]
    if pmc is None and args.pmc != "off":
        try:
            allp = json.load(open(profile_path))
            pmc = (allp.get(args.workload) or {}).get("kernels")
            pmc_source = "committed: profiles/pmc_traffic.json[%s] (tag %s), not collected in this run" % (args.workload, (allp.get(args.workload) or {}).get("_tag"))
        except (OSError, ValueError):
            pmc = None
    work = out.get("gpu_work_per_ray")
    if work is None and args.pmc != "off":
        try:
            work = count_work(args)
            out["gpu_work_per_ray"] = work
        except Exception as e:   # noqa: BLE001
            out["count_work_error"] = str(e)[:300]

    v = (pmc or {}).get(dominant)
    ms_alone, launches, nrays = alone[dominant]
    avg_alone_ms = ms_alone / max(1, launches)
    # the same kernel's average launch in the TIMED region (with overlap: a duration under sharing)
    if dominant.startswith("k_shadow"):
        li_of = [li for li in range(n_lights) if names.get("shadow%d" % li) == dominant]
        timed_ms = sum(sum(st.ms_shadow_light[li] for li in li_of) for st in stats) / steps / max(1, len(li_of) * max(1, r0.launches_closest))
    elif dominant.startswith("k_shade"):
        timed_ms = sum(st.ms_shade for st in stats) / steps / max(1, r0.launches_closest)
    else:
        timed_ms = sum(st.ms_closest - st.ms_shade for st in stats) / steps / max(1, r0.launches_closest)
    roof = {"kernel": dominant, "avg_launch_ms": timed_ms, "avg_launch_ms_alone": avg_alone_ms, "launches_per_frame": launches,
            "rays_per_launch": nrays / max(1, launches), "overlap": not args.serial, "source": pmc_source}
    if v:
        valu_frac = v["busy"] * v["lane_utilisation"]
        hbm_frac = v["hbm_gbs_alone"] / HBM_PEAK_GBS
        bound = "valu" if valu_frac >= hbm_frac else "hbm"
        roof.update({
            "bound": bound, "frac": valu_frac if bound == "valu" else hbm_frac,
            "achieved": valu_frac * VALU_LANE_SLOTS if bound == "valu" else v["hbm_gbs_alone"],
            "peak": VALU_LANE_SLOTS if bound == "valu" else HBM_PEAK_GBS,
            "unit": "VALU lane-slots per cycle doing a lane's work (256 CUs x 4 SIMDs x 16 lanes)" if bound == "valu" else "GB/s",
            "valu_frac": valu_frac, "valu_busy": v["busy"], "lane_utilisation": v["lane_utilisation"],
            "valu_insts_per_wave": v["valu_insts_per_wave"], "waves_waiting_frac": v.get("waves_waiting_frac"),
            "salu_insts_per_wave": v.get("salu_insts_per_wave"), "salu_issue_per_cu_cycle": v.get("salu_issue_per_cu_cycle"),
            "hbm_frac": hbm_frac, "l2_frac": v["l2_request_gbs"] / L2_PEAK_GBS, "hbm_gbs_alone": v["hbm_gbs_alone"],
            "l2_request_gbs": v["l2_request_gbs"], "l2_hit_rate": v.get("l2_hit_rate"),
            "traffic": v["hbm_bytes_per_launch"], "traffic_write": v.get("hbm_write_bytes_per_launch"),
            "avg_launch_ms_alone_profiled": v["alone_ms_per_launch"],
            # the committed profile describes another build if its launch time is far from what this run measures
            "profile_matches_this_run": abs(v["alone_ms_per_launch"] - avg_alone_ms) <= 0.15 * max(avg_alone_ms, 1e-9)})
    else:
        roof.update({"bound": None, "frac": None, "achieved": None, "peak": None, "unit": None, "traffic": None})
    fpr = useful_flops_per_ray(dominant, work)
    if fpr is not None and avg_alone_ms > 0:
        tflops = fpr * (nrays / max(1, launches)) / (avg_alone_ms * 1e-3) / 1e12
        roof.update({"useful_flop_per_ray": fpr, "useful_tflops": tflops, "useful_flop_frac": tflops / FP64_VALU_PEAK_TFLOPS,
                     "useful_flop_peak_tflops": FP64_VALU_PEAK_TFLOPS})
    if cpu is not None:   # SURVEY.md 8d's byte model: reference-order algorithmic bytes the kernels avoid touching; never a roofline fraction
        is_shadow = dominant.startswith("k_shadow")
        bpr = (ocn.algorithmic_bytes(1) / max(1, ocn.rays[4])) if is_shadow else (ocn.algorithmic_bytes(0) / max(1, ocn.rays_closest))
        roof["algorithmic_bytes_per_ray_reference_order"] = bpr
        roof["work_avoidance_vs_reference_order"] = bpr * (nrays / max(1, launches)) / (avg_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["note"] = ("dominant kernel = the largest share of the frame's kernel time, every kernel timed ALONE (one extra RTX_FLAG_SERIAL frame, HIP "
                    "events on the launch's own stream); bound = the larger of valu_frac (VALU busy x lanes active per VALU instruction: issue "
                    "slots doing a lane's work) and hbm_frac (measured HBM bytes / launch time / 8 TB/s); useful_flop_frac = SURVEY 8d flop "
                    "weights x the kernel's own traversal counters (librtx_count.so) / launch time alone / 78.6 TFLOP/s FP64 vector peak: the "
                    "share of the machine doing the ALGORITHM's arithmetic; work_avoidance_vs_reference_order is SURVEY 8d's byte model "
                    "(what the reference's unordered walk would move), > 1 because the kernels do not move those bytes")
    out["roofline"] = roof


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
    mine_ms = torch.tensor([sum(ms_call) / max(1, len(ms_call))], dtype=torch.float64, device=comm_dev)   # this rank's mean frame (call) time
    if world > 1:
        all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)
        per_rank = [float(t.item()) for t in all_ms]
    else:
        per_rank = [float(mine_ms.item())]
    total_rays, elapsed = float(agg[:5].sum()), float(agg[5])
    rounds = math.ceil(len(frames) / world)
    if rank == 0:
        info = scene.info()
        out = {"metric": "Mrays/sec (all ray types)", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
               "steps": rounds, "warmup": args.warmup, "ms_per_step": elapsed / rounds * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s camera animation %g-%g s at %g fps: %d frames of %dx%d, pixel_samples %d (%d spp), shadow_samples %d, "
                                      "reflect %d refract %d, frame k on rank k %% %d, one gather per round of %d frames to rank 0 overlapped with the next round's render; procedural "
                                      "stand-in mesh (reference assets absent), philox seed 0" %
                                      (scene_name, args.time_slice[0], args.time_slice[1], args.fps, len(frames), W, H, ps, ps * ps, shadow,
                                       refl, refr, world, world),
                          "triangles": int(named.n_triangles), "octree_nodes": int(info.n_nodes), "scene_build_upload_s": round(build_s, 3)},
               "frames": len(frames), "frames_rank0": int(n_mine), "ms_per_frame_node": elapsed / len(frames) * 1e3,
               "ms_per_frame_rank0_device": sum(ms_dev) / max(1, len(ms_dev)),
               "ms_per_frame_rank0_call": sum(ms_call) / max(1, len(ms_call)), "frames_per_s_node": len(frames) / elapsed,
               "rays_per_frame": total_rays / len(frames), "last_frame_gathered": last.get("k"),
               "per_rank_frame_ms": {"min": min(per_rank), "max": max(per_rank), "by_rank": [round(x, 3) for x in per_rank]}}
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
