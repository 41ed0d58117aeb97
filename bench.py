#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the batched hot path (48 kHz, block = 512) on MI355X, with the null test beside it.

    python bench.py --gpus N --steps K --warmup W [--leaf DDT] [--instances-total 4096]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: every instance of the job processes 10 s (480 000 frames) of its own
48 kHz white noise in host blocks of 512 (937 full + one 256-frame tail), i.e. one zab_process(frames, block=512) per rank
over that rank's instances, inputs and outputs resident in HBM.

Default workload = the north-star headline (BASELINE.md "headline"): Spatialization/DDT x 4096 instances in
total, STRONG scaling -- rank r of W owns sharding.instance_range(4096, r, W), no data-path collective; RCCL carries the
barrier and the max / sum of the run statistics only. BASELINE.json configs[1] (DDT x 1024 on one GPU) is
`--instances-total 1024`; config C5 is `--leaf ClickBeGoneSG --instances-total 8192`; `--instances-per-gpu K` gives a
weak-scaling run instead. `--gpus N` without a torchrun environment starts the N ranks itself (child torchrun).

One JSON line on rank 0:
  value / ms_per_step .... whole-job throughput of the timed region (max over ranks of the wall time)
  null_test_dbfs ......... max and RMS of (engine output - CPU checker output) in dBFS over a sample of instances of every
                           rank, taken from the FIRST launch of the very engine that is timed (same kernel, same batch
                           shape), outside the timed region. Checker = oracle/port.py (pinned to the reference VM's
                           fixtures) for JSFX leaves, oracle/faust_ref.c for Faust leaves.
  roofline ............... dominant kernel's algorithmic HBM bytes / its mean duration (HIP events on the engine stream);
                           `traffic` is read from the committed PMC passes of this command (`traffic_source` says which)
  cpu_baseline ........... the CPU checker of the same leaf timed on this box's host cores on a bounded sample (kind "port"),
                           plus `reference_vm`: the reference's own WDL/EEL2 VM (oracle/_ref, built from the reference
                           sources) timed on one core on the repo-authored tap-delay script tests/fixtures/delaytaps.jsfx
                           (the reference's leaf scripts do not travel to the GPU box), with the port's speed on the same
                           script beside it.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
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
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~5 TB/s is what a device copy reaches
MIN_TIMED_S = 2.5         # --steps default: as many as make the timed region at least this long


# ----------------------------------------------------------------------------------------------------------------------
# CPU checker leg (the only part of this file that touches oracle/): null test + cpu_baseline
# ----------------------------------------------------------------------------------------------------------------------
class Checker:
    """CPU restatement of one leaf: oracle/port.py for JSFX leaves, oracle/faust_ref.c for Faust leaves."""

    def __init__(self, leaf: str, meta: dict):
        self.leaf, self.meta = leaf, meta
        self.faust = meta.get("kind") == "faust"
        self.kind_text = ("oracle/faust_ref.c (CPU restatement of the .dsp, gcc -O2, scalar f32; parity unpinned)" if self.faust
                          else "oracle/port.py (AOT lowering compiled by g++ -O2, scalar f64; pinned to reference-VM fixtures)")

    def fresh(self):
        if self.faust:
            from oracle import faust_ref
            return faust_ref.FaustRef(self.leaf, SRATE)
        from oracle import port
        p = port.Port(self.leaf, SRATE)
        p.set_sliders(self.meta["default_sliders"])
        p.prepare()
        return p

    def run(self, obj, x, block):
        if self.faust:
            return obj.compute(x, self.meta["default_sliders"][:8], block=block)
        return obj.process(x, block)


def null_test(checker: Checker, outs: dict):
    """outs: {instance: (x, y)} float32 [nch, frames] input (as generated in HBM) and output of the engine's first launch.
    Returns (max |d|, sum d^2, count)."""
    mx, ss, cnt = 0.0, 0.0, 0
    for _, (x, y) in outs.items():
        ref = checker.run(checker.fresh(), x, BLOCK)
        d = y.astype(np.float64) - ref.astype(np.float64)
        mx = max(mx, float(np.abs(d).max()))
        ss += float((d * d).sum())
        cnt += d.size
    return mx, ss, cnt


def cpu_baseline(checker: Checker, nch: int, seconds_budget: float = 10.0):
    """The leaf's CPU checker on this box's host cores, bounded sample: one instance per thread (ctypes releases the GIL),
    2 s passes of the benchmark's own noise repeated for ~seconds_budget."""
    from zajit import noise
    cores = max(1, min(os.cpu_count() or 1, 16))
    frames = 96_000
    x = noise.white_noise(range(cores), frames, channels=nch)

    def work(i, budget=seconds_budget):
        obj = checker.fresh()
        t = time.perf_counter()
        reps = 0
        while True:
            checker.run(obj, x[i], BLOCK)
            reps += 1
            if time.perf_counter() - t > budget:
                break
        return reps

    t1 = time.perf_counter()
    one = work(0, 2.5) * frames * nch / (time.perf_counter() - t1) / 1e6
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        reps = list(ex.map(work, range(cores)))
    wall = time.perf_counter() - t0
    return {"value": sum(reps) * frames * nch / wall / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "one_core": one, "checker": checker.kind_text,
            "sample": f"{checker.leaf} defaults, {cores} threads x 1 instance, {frames}-frame passes repeated for ~{seconds_budget:.0f} s, block {BLOCK}"}


def reference_vm_baseline(seconds_budget: float = 4.0):
    """The reference's own CPU path -- its WDL/EEL2 VM, compiled from the reference sources into oracle/_ref -- on one host
    core, on the repo-authored tap-delay script (the leaf scripts themselves stay in the reference tree), and the CPU port
    on the same script: the ratio relates `cpu_baseline.value` (port) to what the reference VM would do."""
    from oracle import eel_oracle, port
    from zajit import noise, program, sliders
    src = ROOT / "tests" / "fixtures" / "delaytaps.jsfx"
    if not eel_oracle.available() or not src.exists():
        return None
    text = program.expand_imports(src)
    prog = program.analyse(text, "fx_delaytaps")
    row = sliders.default_slider_values(prog.slider_decls)
    frames = 24_000
    x = noise.white_noise([0], frames)[0]
    vm = eel_oracle.EelOracle(text, prog.aliases)
    vm.set_sliders(row)
    vm.prepare(SRATE)
    vm.set_write_trace(False)        # the correctness monitor's store instrumentation is not part of the VM's own speed
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds_budget:
        vm.process(x, BLOCK)
        reps += 1
    vm_rate = reps * frames * 2 / (time.perf_counter() - t0) / 1e6
    out = {"value": vm_rate, "unit": "Msamples/s", "cores": 1, "kind": "reference VM on a repo-authored script of the leaf's class, not the leaf itself",
           "sample": f"WDL/EEL2 portable VM (oracle/_ref, gcc -O2) on tests/fixtures/delaytaps.jsfx defaults, {frames}-frame passes "
                     f"for ~{seconds_budget:.0f} s, block {BLOCK}, one core"}
    if port.port_path("fx_delaytaps").exists():
        p = port.Port("fx_delaytaps", SRATE)
        p.set_sliders(row)
        p.prepare()
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.5:
            p.process(x, BLOCK)
            reps += 1
        out["port_same_script_one_core"] = reps * frames * 2 / (time.perf_counter() - t0) / 1e6
    return out


# ----------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(leaf: str, n_inst: int, frames: int, nch_io: int, nvars: int, ddt_state, mem_words: int) -> float:
    """the kernels are persistent over a launch (state stays on-chip across host blocks), so per frame
    only the audio moves: 4 B in + 4 B out per channel. Once per launch and instance: vars r/w, and
      DDT:      delay history read 2 x 8 x H, ring write-back 2 x 8 x min(frames, 16384), tap tables 5 x 8 x tapN
      others:   the arena footprint (write high-water mark) read + written once."""
    per_frame = 8.0 * nch_io
    once = 2 * 8 * nvars
    if ddt_state:
        once += 2 * 8 * ddt_state["H"] + 2 * 8 * min(frames, 16384) + 5 * 8 * ddt_state["tapN"]
    else:
        once += 2 * 8 * mem_words
    return n_inst * (per_frame * frames + once)


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torchrun (nothing here has touched the GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return subprocess.call(cmd)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help=f"0 = as many as make the timed region >= {MIN_TIMED_S} s")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--leaf", default="DDT")
    ap.add_argument("--instances-total", type=int, default=4096, help="strong scaling: split over the ranks")
    ap.add_argument("--instances-per-gpu", type=int, default=0, help="weak scaling: this many on every rank")
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--path", choices=["auto", "generic", "fast"], default="auto")
    ap.add_argument("--null-instances", type=int, default=4, help="instances per rank checked against the CPU checker")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-null-test", action="store_true")
    ap.add_argument("--mem-cap", type=int, default=0, help="mem[] cells per instance (0: the leaf's default arena of 65536; Alias needs 524288)")
    ap.add_argument("--group", action="store_true",
                    help="ONE process over the C library's zab_group_* (a host thread + stream per GPU, RCCL for the end-of-run "
                         "statistics) instead of one torchrun rank per GPU; same shards, same JSON line")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.group:
        if world != 1:
            print("bench.py: --group is a single process (do not start it under torchrun)", file=sys.stderr)
            return 2
        world = args.gpus
    elif args.gpus != world:
        if world == 1 and "RANK" not in os.environ and args.gpus > 1:
            return spawn_ranks(args.gpus, sys.argv[1:])
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}", file=sys.stderr)
        return 2

    import torch
    import sharding
    import zabatch

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the engine has no CPU path)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    if args.group and torch.cuda.device_count() < args.gpus:
        print(f"bench.py: --group --gpus {args.gpus} but {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        return 2
    if world > 1 and not args.group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    leaf = args.leaf
    meta = zabatch.leaf_meta(leaf)
    try:
        if args.group:          # this process holds every shard (zab_group_create lays them out as sharding.plan does over ranks)
            for r in range(world):
                sharding.plan(r, world, args.instances_total, args.instances_per_gpu)
            shard = sharding.Shard(0, args.instances_per_gpu * world or args.instances_total, args.instances_per_gpu * world or args.instances_total,
                                   "weak" if args.instances_per_gpu else "strong")
        else:
            shard = sharding.plan(rank, world, args.instances_total, args.instances_per_gpu)
    except ValueError as ex:
        print(f"bench.py: {ex}", file=sys.stderr)
        return 2
    weak, lo, hi, n_total, n_inst = shard.scaling == "weak", shard.lo, shard.hi, shard.n_total, shard.count
    frames = args.frames
    path = {"auto": zabatch.ZAB_PATH_AUTO, "generic": zabatch.ZAB_PATH_GENERIC, "fast": zabatch.ZAB_PATH_FAST}[args.path]
    grp = None
    if args.group:
        # the whole job in this process: zab_group_create shards n_total instances over the devices exactly as sharding.plan
        # does over ranks (contiguous ranges); shard 0's engine stands where rank 0's engine stands below
        grp = zabatch.Group(leaf, n_total, devices=list(range(world)), srate=SRATE, max_block=BLOCK, path=path, first_instance_id=1,
                            **({"mem_cap": args.mem_cap} if args.mem_cap else {}))
        grp.set_sliders(meta["default_sliders"])
        grp.prepare()
        eng = grp.shards[0][2]
        nch = eng.nch
        n_inst = grp.shards[0][1]
        g_in, g_out = [], []
        for first, count, view in grp.shards:
            nb = count * nch * frames * 4
            g_in.append(view.device_alloc(nb)); g_out.append(view.device_alloc(nb))
            view.device_noise(g_in[-1], frames, id_offset=first)
        grp.sync()
        d_in, d_out = g_in[0], g_out[0]
    else:
        eng = zabatch.Engine(leaf, n_inst, srate=SRATE, max_block=BLOCK, device=local_rank, path=path, first_instance_id=1 + lo,
                             mem_cap=args.mem_cap)
        nch = eng.nch
        eng.set_sliders(meta["default_sliders"])
        try:
            eng.prepare()
        except zabatch.ZabError as ex:
            # the leaf's @init stores past the default arena (a fixed arena reports what it would have needed: DESIGN.md section 3):
            # size it from the report, once, unless the caller chose a size
            import re as _re
            m = _re.search(r"needed >= (\d+)", str(ex))
            if args.mem_cap or not m:
                raise
            eng.close()
            args.mem_cap = 1 << (int(m.group(1)) + 64).bit_length()
            print(f"bench.py: {leaf} needs an arena of {int(m.group(1))} cells: mem_cap = {args.mem_cap}", file=sys.stderr)
            eng = zabatch.Engine(leaf, n_inst, srate=SRATE, max_block=BLOCK, device=local_rank, path=path, first_instance_id=1 + lo,
                                 mem_cap=args.mem_cap)
            eng.set_sliders(meta["default_sliders"])
            eng.prepare()
        nbytes = n_inst * nch * frames * 4
        d_in, d_out = eng.device_alloc(nbytes), eng.device_alloc(nbytes)
        eng.device_noise(d_in, frames, id_offset=lo)      # synthetic white noise, generated in HBM; noise id = global instance index
        eng.sync()

    def launch():
        if grp is not None:
            grp.process_device(g_in, g_out, frames, block=BLOCK)
        else:
            eng.process_device(d_in, d_out, frames, block=BLOCK)

    def sync_all():
        if grp is not None:
            grp.sync()
        else:
            eng.sync()

    def barrier():
        sync_all()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # first launch of the timed engine: its output for a sample of instances is what the null test checks
    launch()
    sync_all()
    t_first = eng.last_timing()[0] * 1e-3
    sample_out = {}
    if not args.no_null_test:
        # (a null test at EVERY shard: the waves per instance -- hence the last bits -- follow the shard's size, d2_pick_nw)
        for first, count, view, bi, bo in ([(lo, n_inst, eng, d_in, d_out)] if grp is None else
                                           [(f, c, v, g_in[k_], g_out[k_]) for k_, (f, c, v) in enumerate(grp.shards)]):
            k = max(1, min(args.null_instances if grp is None else max(1, args.null_instances // world), count))
            for j in sorted({int(round(q * (count - 1) / max(1, k - 1))) for q in range(k)}):
                off = j * nch * frames * 4
                sample_out[first + j] = (view.download(bi + off, (nch, frames)), view.download(bo + off, (nch, frames)))
    for _ in range(max(0, args.warmup - 1)):
        launch()
    sync_all()
    t_step = eng.last_timing()[0] * 1e-3
    steps = args.steps
    if steps <= 0:
        steps = max(10, int(math.ceil(MIN_TIMED_S / max(t_step, 1e-6))))
        if dist is not None:
            t = torch.tensor([steps], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            steps = int(t[0])
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        launch()
    sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    # device duration of each launch of the timed region: HIP event pairs recorded on the engine's own stream
    kernel_ms = eng.timing_history(min(steps, 64))
    used_fast = eng.used_fast_path()
    kernel_name = eng.last_kernel_name()

    checker = Checker(leaf, meta)
    null_mx, null_ss, null_n = (0.0, 0.0, 0)
    if sample_out:
        null_mx, null_ss, null_n = null_test(checker, sample_out)
    units_here = float(n_total if grp is not None else n_inst) * nch * frames * steps
    job = sharding.reduce_stats(sharding.RunStats(elapsed_s=elapsed, units=units_here, max_abs_err=null_mx),
                                dist, device="cuda" if dist is not None else None)
    group_stats = grp.reduce() if grp is not None else None      # (max / sum of the shards' kernel times: RCCL when they sit on several GPUs)
    if dist is not None:
        t = torch.tensor([null_ss, float(null_n)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        null_ss, null_n = float(t[0]), int(t[1])
    elapsed = job.elapsed_s            # MAX over ranks

    # state-dependent constants for the algorithmic byte count
    names = eng.var_names()
    ddt_state = None
    if leaf == "DDT":
        v = eng.read_vars(0, 1)[0]
        tapN = int(v[names.index("tapN")])
        dl = eng.read_mem(32768, 64, 0, 1)[0][:tapN]
        dr = eng.read_mem(32768 + 64, 64, 0, 1)[0][:tapN]
        # history the taps can reach back into = Dmax frames (the kernel stages a longer ring; the surplus is not counted)
        ddt_state = {"H": int(max(dl.max(), dr.max())), "tapN": tapN}
    mem_words = int(eng.mem_high(0, 1)[0]) if meta.get("kind") != "faust" else 0
    nch_io = (meta["io"]["inputs"] + meta["io"]["outputs"]) / 2.0 if "io" in meta else nch
    alg = algorithmic_bytes(leaf, n_inst, frames, nch_io, len(names), ddt_state, mem_words)

    if rank == 0:
        total_samples = job.units      # SUM over ranks
        k_ms = float(np.mean(kernel_ms))
        achieved = alg / (k_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: PMC counters need the profiler around the process, so they cannot be read inside
        # this run; tools/pmc_pass.sh runs THIS command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes)
        # and leaves profiles/pmc_<kernel>_<instances>x<frames>.json, which is picked up here by kernel name and batch shape
        # (profiles/pmc_traffic.json: the default configuration's, kept under its old name too)
        traffic, traffic_source = None, None
        kname = kernel_name
        for pmc in (ROOT / "profiles" / f"pmc_{kname}_{n_inst}x{frames}.json", ROOT / "profiles" / "pmc_traffic.json"):
            if traffic is not None or not pmc.exists():
                continue
            try:
                rec = json.loads(pmc.read_text())
                if (rec.get("leaf", "DDT") == leaf and rec.get("instances") == n_inst and rec.get("frames") == frames
                        and rec.get("fast") == used_fast and rec.get("kernel", kname) == kname):
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = (f"profiles/{pmc.name}: committed rocprofv3 --pmc passes of this command "
                                      "(FETCH_SIZE / WRITE_SIZE, separate runs; tools/pmc_pass.sh), not measured in this run")
            except Exception:
                traffic = None
        null_db = lambda a: None if null_n == 0 else (float(20.0 * math.log10(a)) if a > 0 else -400.0)
        line = {
            "metric": "Msamples/sec across batched instances, 48 kHz stereo block=512; null-test dBFS",
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": max(1, args.warmup),
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32" if meta.get("kind") == "faust" else "f64",
            "data": "synthetic",
            "null_test_dbfs": null_db(job.max_abs_err),
            "null_test": {"max_dbfs": null_db(job.max_abs_err), "rms_dbfs": null_db(math.sqrt(null_ss / null_n)) if null_n else None,
                          "max_abs": job.max_abs_err if null_n else None, "instances_checked": null_n // max(1, nch * frames),
                          "frames": frames, "pass_bar_dbfs": -100.0, "checker": checker.kind_text,
                          "what": "first launch of the timed engine vs the CPU checker on the same noise, outside the timed region"},
            "config": {"workload": f"{leaf} x{n_total} instances in total ({'weak' if weak else 'strong'} scaling, "
                                   f"{n_inst} on rank 0), defaults, 48 kHz x{nch} ch, {frames} frames white noise, block={BLOCK}",
                       "leaf": leaf, "instances_total": n_total, "instances_rank0": n_inst, "frames_per_step": frames,
                       "kernel": kernel_name,
                       "sharding": (f"zab_group_* in one process: {world} shard(s), a host thread + stream per GPU, no data-path collective"
                                    if grp is not None else f"sharding.instance_range over {world} rank(s), no data-path collective"),
                       "driver": "group" if grp is not None else "torchrun"},
            "mframes_per_s": total_samples / nch / elapsed / 1e6,
            "realtime_factor_per_instance": frames / SRATE / (elapsed / steps),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms": k_ms, "kernel_ms_min": float(np.min(kernel_ms)),
                         "kernel_ms_first_launch": t_first * 1e3,
                         "algorithmic_bytes_per_launch": alg, "bytes_per_frame": alg / (n_inst * frames)},
        }
        if ddt_state:
            # context: the script's own arithmetic, 14 flop per tap + 80 per frame, all f64, against the FP64
            # vector peak (half the guide's 157.3 TFLOP/s FP32 vector figure)
            fl = 14 * ddt_state["tapN"] + 80
            line["roofline"].update(fp64_flop_per_frame=fl, fp64_tflops=fl * n_inst * frames / (k_ms * 1e-3) / 1e12,
                                    fp64_vector_peak_tflops=78.6)
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
        if group_stats is not None:
            line["group"] = {k: group_stats[k] for k in ("n_shards", "used_rccl", "max_kernel_ms", "sum_kernel_ms")}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(checker, nch)
            try:
                ref = reference_vm_baseline()
            except Exception as ex:  # noqa: BLE001  (the VM library is optional on a box that never saw the reference)
                ref = {"error": str(ex)[:200]}
            if ref:
                line["cpu_baseline"]["reference_vm"] = ref
        print(json.dumps(line), flush=True)
    if grp is not None:
        grp.close()
    else:
        eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
