#!/usr/bin/env python3
"""The collectives of bench.py's N > 1 flow on a group of ONE rank over RCCL (a one-GPU box has no second card: tools/gpu_mgpu_rehearsal.sh
runs two ranks over gloo, this runs the `nccl` calls themselves): init_process_group("nccl", device_id=), the still's gather of compact
8-row blocks (StillGather), the animation's pipelined gather with the flag in the trailer (FrameGather through sharding.animate), the
all_reduce on a slice, all_gather, barrier, destroy.  The gathered frames must equal the frames rendered without any collective.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/gpu_rccl_rehearsal.py
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == 1 and rank == 0, "a rehearsal for one rank"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev).cpu()
    t0 = time.time()
    dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    init_s = time.time() - t0

    pkg = importlib.import_module("rust-raytracer_amd")
    sh = pkg.sharding
    W, H = 640, 360
    named = pkg.host.NamedScene("sibenik", detail=0.25)
    scene = pkg.Scene(named, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"backend": dist.get_backend(), "init_s": round(init_s, 3)}

    # ---- still: compact blocks, one gather, de-interleave (bench.py main)
    cam = named.camera(W, H, 60.0)
    opt = pkg.options(reflect_depth=2, refract_depth=2, shadow_samples=4, gloss_samples=1, pixel_samples=1, rng_mode=pkg.abi.RNG_PHILOX,
                      seed=0, frame=0, shard_index=rank, shard_count=world, flags=pkg.abi.FLAG_COMPACT_OUTPUT)
    gat = sh.StillGather(W, H, rank, world, dev, always_collective=True)
    for _ in range(3):
        scene.render_device(cam, opt, gat.target.data_ptr(), None, stream)
        final = gat.gather(dist)
    dist.barrier()
    torch.cuda.synchronize()
    plain = sh.StillGather(W, H, rank, world, dev)
    scene.render_device(cam, opt, plain.target.data_ptr(), None, stream)
    ref = plain.gather(None)
    torch.cuda.synchronize()
    out["still_equal"] = bool(torch.equal(final, ref))
    out["still_nonzero"] = int((final[..., :3] != 0).sum().item())

    # ---- the reductions of the bench line
    agg = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 0.25], dtype=torch.float64, device=dev)
    tmax = agg[5:].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(agg[:5], op=dist.ReduceOp.SUM)
    agg[5] = tmax[0]
    mine = torch.tensor([7.5], dtype=torch.float64, device=dev)
    all_ms = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(all_ms, mine)
    out["reductions_ok"] = agg.cpu().tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 0.25] and all_ms[0].item() == 7.5

    # ---- animation: frames through the pipelined gather (bench.py animate)
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    frames = sh.animation_frames(25.0, 0.0, 0.2)     # 5 frames
    sums_direct, sums_gathered = [], []

    def render_frame(k, t):
        c = named.camera(W, H, 60.0, time=t)
        o = pkg.options(reflect_depth=2, refract_depth=2, shadow_samples=4, gloss_samples=1, pixel_samples=1, rng_mode=pkg.abi.RNG_PHILOX,
                        seed=0, frame=k)
        scene.render_device(c, o, rgba.data_ptr(), None, stream)
        sums_direct.append(int(rgba.sum(dtype=torch.int64).item()))
        return rgba

    def on_frame(k, t, frame):
        sums_gathered.append((k, int(frame.sum(dtype=torch.int64).item())))

    n = sh.animate(frames, rank, world, dist, render_frame, on_frame, always_collective=True)
    dist.barrier()
    torch.cuda.synchronize()
    out["anim_frames"] = n
    out["anim_order"] = [k for k, _ in sums_gathered] == [k for k, _ in frames]
    out["anim_equal"] = [s for _, s in sums_gathered] == sums_direct
    scene.close()
    dist.barrier()
    dist.destroy_process_group()
    out["ok"] = bool(out["still_equal"] and out["still_nonzero"] > 0 and out["reductions_ok"] and out["anim_order"] and out["anim_equal"]
                     and n == len(frames))
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
