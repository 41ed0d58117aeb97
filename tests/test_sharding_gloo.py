"""N>1 path on CPU: two processes over gloo shard a batch, each runs its shard, statistics reduce correctly.
The per-rank engine is replaced here by the CPU port (no GPU in this suite); what is under test is the sharding and the
control-plane reduction that bench.py uses with the nccl (RCCL) backend on GPUs."""
import os
import socket

import numpy as np
import pytest


def test_instance_ranges_partition_exactly():
    import sharding
    for n in (0, 1, 7, 8, 1024, 4097):
        for w in (1, 2, 3, 8):
            spans = [sharding.instance_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            for i in range(0, n, max(1, n // 13)):
                r = sharding.owner_of(i, n, w)
                assert spans[r][0] <= i < spans[r][1]
    with pytest.raises(ValueError):
        sharding.instance_range(4, 2, 2)


def test_shard_plans():
    import sharding
    assert [sharding.plan(r, 2, instances_total=5).count for r in range(2)] == [3, 2]          # uneven: earlier ranks get the extra
    assert [(s.lo, s.hi) for s in (sharding.plan(r, 3, instances_per_rank=4) for r in range(3))] == [(0, 4), (4, 8), (8, 12)]
    assert sharding.plan(1, 3, instances_per_rank=4).scaling == "weak" and sharding.plan(1, 3, instances_per_rank=4).n_total == 12
    with pytest.raises(ValueError):
        sharding.plan(3, 4, instances_total=3)                                                 # a rank with nothing to do


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, frames, out_dir):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for p in (str(root / "zorakaudio-experimental-plugins_amd"), str(root)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import sharding
    import zabatch
    from oracle import port as cpu_port
    from zajit import noise
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = sharding.plan(rank, world, instances_total=n_total)          # what bench.py does with --instances-total
    lo, hi = shard.lo, shard.hi
    assert shard.scaling == "strong" and shard.n_total == n_total and (lo, hi) == sharding.instance_range(n_total, rank, world)
    meta = zabatch.leaf_meta("DDT")
    x = noise.white_noise(range(lo, hi), frames)
    y = np.zeros_like(x)
    for j in range(hi - lo):
        p = cpu_port.Port("DDT", 48000.0)
        p.set_sliders(meta["default_sliders"]); p.prepare()
        y[j] = p.process(x[j], 512)
    np.save(os.path.join(out_dir, f"y_{rank}.npy"), y)
    dist.barrier()
    st = sharding.reduce_stats(sharding.RunStats(elapsed_s=1.0 + rank, units=float((hi - lo) * 2 * frames),
                                                 max_abs_err=1e-9 * (rank + 1)), dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "stats.npy"), np.array([st.elapsed_s, st.units, st.max_abs_err]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    import zabatch
    from oracle import port as cpu_port
    from zajit import noise
    if not cpu_port.port_path("DDT").exists() or not (zabatch.LIB_DIR / "DDT.json").exists():
        pytest.skip("DDT port / metadata not built")
    n_total, frames, world = 5, 700, 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, frames, str(tmp_path)), nprocs=world, join=True)
    y = np.concatenate([np.load(tmp_path / f"y_{r}.npy") for r in range(world)], axis=0)
    # the sharded run equals the unsharded one, instance for instance
    meta = zabatch.leaf_meta("DDT")
    x = noise.white_noise(range(n_total), frames)
    for i in range(n_total):
        p = cpu_port.Port("DDT", 48000.0)
        p.set_sliders(meta["default_sliders"]); p.prepare()
        assert np.array_equal(p.process(x[i], 512), y[i])
    elapsed, units, err = np.load(tmp_path / "stats.npy")
    assert elapsed == 2.0 and units == n_total * 2 * frames and abs(err - 2e-9) < 1e-18
