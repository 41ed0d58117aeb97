#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the batched DDT hot path (48 kHz stereo, block = 512) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: every instance on this rank processes 10 s (480 000 frames) of
its own 48 kHz stereo white noise in host blocks of 512 (937 full + one 256-frame tail), i.e. one
zab_process(frames=480000, block=512) over N_inst instances with inputs and outputs resident in HBM.
Workload at N=1 = BASELINE.json configs[1]: "DDT 1024 batched instances on 1 MI355X, block=512".
Instances are independent, so ranks shard them with no data-path collective (weak scaling: 1024 instances per GPU);
RCCL is used only for the barrier and the max-over-ranks of the timed region.

One JSON line on rank 0 (see README/DESIGN for the field definitions), including
  roofline     -- dominant kernel's algorithmic HBM bytes / its mean duration (HIP events on the engine stream)
  cpu_baseline -- the CPU port of the same path (oracle/port.py, g++ -O2 scalar f64; kind "port") timed on this box's host
                  cores on a bounded sample. (The reference's EEL2 VM needs the leaf's script text, which does not travel
                  to the GPU box; its speed is recorded in DESIGN.md.)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
PKG = ROOT / "zorakaudio-experimental-plugins_amd"
for p in (str(PKG), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

SRATE = 48000.0
FRAMES = 480_000          # 10 s
BLOCK = 512
NCH = 2
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel achieves


def algorithmic_bytes_per_launch(n_inst: int, frames: int, meta_state: dict) -> float:
    """DESIGN.md §roofline. The DDT kernel is persistent over the launch (state stays on-chip across host blocks), so
    per frame only the audio moves: 2 ch x 4 B in + 2 ch x 4 B out = 16 B. Per launch and instance, once:
    delay history read 2 x 8 x H, ring write-back 2 x 8 x min(frames, 16384), tap tables 5 x 8 x tapN, vars r/w."""
    per_frame = 16.0
    H, tapN, nvars = meta_state["H"], meta_state["tapN"], meta_state["nvars"]
    per_launch = 2 * 8 * H + 2 * 8 * min(frames, 16384) + 5 * 8 * tapN + 2 * 8 * nvars
    return n_inst * (per_frame * frames + per_launch)


def cpu_baseline(seconds_budget: float = 12.0):
    """CPU restatement of the same hot path (oracle/port.py: the AOT lowering compiled by g++ -O2, scalar f64 -- the
    stand-in for the reference's LLVM-AOT object) on this box's host cores, on a bounded sample of the same workload:
    DDT defaults, 48 kHz stereo white noise, block 512, one instance per thread (ctypes releases the GIL).
    The reference's own EEL2 VM cannot serve here: it needs the leaf's script text, which must not travel to the GPU
    box; its speed measured in the dev container is recorded in DESIGN.md instead."""
    from oracle import port
    import zabatch
    from zajit import noise
    meta = zabatch.leaf_meta("DDT")
    cores = max(1, min(os.cpu_count() or 1, 16))
    frames = 96_000   # 2 s of audio per pass and thread
    x = noise.white_noise(range(cores), frames)

    def work(i, budget=seconds_budget):
        p = port.Port("DDT", SRATE)
        p.set_sliders(meta["default_sliders"]); p.prepare()
        t = time.perf_counter()
        reps = 0
        while True:
            p.process(x[i], BLOCK)
            reps += 1
            if time.perf_counter() - t > budget:
                break
        return reps

    t1 = time.perf_counter()
    one = work(0, 3.0) * frames * NCH / (time.perf_counter() - t1) / 1e6       # 1 core beside all cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        reps = list(ex.map(work, range(cores)))
    wall = time.perf_counter() - t0
    total = sum(reps) * frames * NCH
    return {"value": total / wall / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "one_core": one,
            "sample": f"DDT defaults, {cores} threads x 1 instance, {frames}-frame passes repeated for ~{seconds_budget:.0f} s, block 512, g++ -O2 scalar f64"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--instances-per-gpu", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--path", choices=["auto", "generic", "fast"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import zabatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the engine has no CPU path)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_inst = args.instances_per_gpu
    frames = args.frames
    path = {"auto": zabatch.ZAB_PATH_AUTO, "generic": zabatch.ZAB_PATH_GENERIC, "fast": zabatch.ZAB_PATH_FAST}[args.path]
    meta = zabatch.leaf_meta("DDT")
    eng = zabatch.Engine("DDT", n_inst, srate=SRATE, max_block=BLOCK, device=local_rank, path=path,
                         first_instance_id=1 + rank * n_inst)
    eng.set_sliders(meta["default_sliders"])
    eng.prepare()
    nbytes = n_inst * NCH * frames * 4
    d_in, d_out = eng.device_alloc(nbytes), eng.device_alloc(nbytes)
    eng.device_noise(d_in, frames, id_offset=rank * n_inst)      # synthetic white noise, generated in HBM
    eng.sync()

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        eng.process_device(d_in, d_out, frames, block=BLOCK)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.process_device(d_in, d_out, frames, block=BLOCK)
    eng.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    # device duration of each launch of the timed region: HIP event pairs recorded on the engine's own stream
    kernel_ms = eng.timing_history(min(args.steps, 64))
    used_fast = eng.used_fast_path()
    kernel_name = eng.last_kernel_name()
    import sharding
    job = sharding.reduce_stats(sharding.RunStats(elapsed_s=elapsed, units=float(n_inst) * NCH * frames * args.steps),
                                dist, device="cuda" if dist is not None else None)
    elapsed = job.elapsed_s            # MAX over ranks

    # state-dependent constants for the algorithmic byte count
    names = eng.var_names()
    v = eng.read_vars(0, 1)[0]
    tapN = int(v[names.index("tapN")])
    dl = eng.read_mem(32768, 64, 0, 1)[0][:tapN]
    dr = eng.read_mem(32768 + 64, 64, 0, 1)[0][:tapN]
    dmax = int(max(dl.max(), dr.max()))
    # history the taps can reach back into = Dmax frames (the kernel stages a power-of-two ring; the surplus is not counted)
    alg = algorithmic_bytes_per_launch(n_inst, frames, {"H": dmax, "tapN": tapN, "nvars": len(names)})

    if rank == 0:
        total_samples = job.units      # SUM over ranks
        k_ms = float(np.mean(kernel_ms))
        achieved = alg / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                rec = json.loads(pmc.read_text())
                if rec.get("instances") == n_inst and rec.get("frames") == frames and rec.get("fast") == used_fast:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/sec across batched instances, 48 kHz stereo block=512",
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"DDT x{n_inst} instances per GPU, defaults, 48 kHz stereo, {frames} frames white noise, block={BLOCK}",
                       "leaf": "Spatialization/DDT", "instances_total": world * n_inst, "frames_per_step": frames,
                       "kernel": kernel_name, "sharding": f"instances x{world}, no collective"},
            "mframes_per_s": total_samples / NCH / elapsed / 1e6,
            "realtime_factor_per_instance": frames / SRATE / (elapsed / args.steps),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg, "bytes_per_frame": alg / (n_inst * frames),
                         # context: the script's own arithmetic, 14 flop per tap + 80 per frame, all f64,
                         # against the FP64 vector peak (half the guide's 157.3 TFLOP/s FP32 vector figure)
                         "fp64_flop_per_frame": 14 * tapN + 80,
                         "fp64_tflops": (14 * tapN + 80) * n_inst * frames / (k_ms * 1e-3) / 1e12,
                         "fp64_vector_peak_tflops": 78.6},
        }
        line["roofline"]["fp64_frac"] = line["roofline"]["fp64_tflops"] / 78.6
        try:    # what a plain device copy reaches on this box (SURVEY §8d asks for it beside the spec peak)
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")        # 1 GiB
            dst = torch.empty_like(src)
            dst.copy_(src); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record(); torch.cuda.synchronize()
            copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            line["roofline"]["device_copy_gbs"] = copy_gbs
            line["roofline"]["frac_of_device_copy"] = achieved / copy_gbs
            del src, dst
        except Exception:
            pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
