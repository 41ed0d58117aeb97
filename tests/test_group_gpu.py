"""Multi-GPU host API of the C library (zab_group_*): a job sharded by instance, one engine / stream / host thread per
shard, no collective on the data path, RCCL for the end-of-run statistics. A one-GPU box can still exercise all of it: two shards
on the same device cover the uneven split (5 instances -> 3 + 2), the per-shard threads and the slider routing; a one-shard group
goes through the RCCL communicator (a single rank)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("leaf", ["DDT", "fx_dynkat"])
def test_group_of_two_shards_equals_one_engine(leaf):
    import zabatch
    from zajit import noise
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    n, frames = 5, 3000
    x = noise.white_noise(range(n), frames)
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    first = min(int(k) for k in meta["sliders"])
    sd = meta["sliders"][str(first)]
    rows[:, first] = np.linspace(sd["min"], sd["max"], n + 2)[1:-1]               # every instance its own setting
    with zabatch.Engine(leaf, n) as e:
        e.set_sliders(rows); e.prepare()
        want = e.process_host(x, block=512)
        want_v = e.read_vars()
    with zabatch.Group(leaf, n, devices=[0, 0]) as g:
        assert [(f, c) for f, c, _ in g.shards] == [(0, 3), (3, 2)]
        g.set_sliders(rows)                                                      # routed to the shards by global instance number
        g.prepare()
        got = g.process_host(x, block=512)
        got_v = np.concatenate([ev.read_vars() for _, _, ev in g.shards], axis=0)
        st = g.reduce(shard_values=[1e-9, 3e-9])
        g.set_sliders(rows[3:4] * 0 + rows[0], first=3, count=1)                 # instance 3 (shard 1, local 0) takes instance 0's row
        again = g.process_host(x[:, :, :512], block=512)
    assert np.array_equal(got, want) and np.array_equal(got_v, want_v)
    assert st["n_shards"] == 2 and st["used_rccl"] == 0 and st["max_value"] == 3e-9
    assert st["units"] == n * 2 * frames and st["max_kernel_ms"] > 0 and st["sum_kernel_ms"] >= st["max_kernel_ms"]
    assert np.isfinite(again).all()


def test_single_shard_group_reduces_over_rccl():
    import zabatch
    from zajit import noise
    n, frames = 4, 1024
    x = noise.white_noise(range(n), frames)
    meta = zabatch.leaf_meta("DDT")
    with zabatch.Group("DDT", n, devices=[0]) as g:
        g.set_sliders(meta["default_sliders"]); g.prepare()
        y = g.process_host(x, block=512)
        st = g.reduce(shard_values=[2.5e-7])
    with zabatch.Engine("DDT", n) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        assert np.array_equal(e.process_host(x, block=512), y)
    assert st["used_rccl"] == 1 and st["n_shards"] == 1 and st["max_value"] == 2.5e-7 and st["units"] == n * 2 * frames


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("leaf", ["DDT", "fx_stft"])
def test_group_on_two_devices_equals_one_engine(leaf):
    """The case a one-GPU box cannot run: shards on two DISTINCT devices -- their own
    allocations, streams and host threads, the end-of-run statistics reduced over RCCL between them -- against one engine on
    device 0. Skipped where fewer than two devices are visible."""
    import zabatch
    from zajit import noise
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    n, frames = 7, 6000
    x = noise.white_noise(range(n), frames)
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    first = min(int(k) for k in meta["sliders"])
    sd = meta["sliders"][str(first)]
    rows[:, first] = np.linspace(sd["min"], sd["max"], n + 2)[1:-1]
    with zabatch.Engine(leaf, n) as e:
        e.set_sliders(rows); e.prepare()
        want = e.process_host(x, block=512)
        want_v = e.read_vars()
    with zabatch.Group(leaf, n, devices=[0, 1]) as g:
        assert [(f, c) for f, c, _ in g.shards] == [(0, 4), (4, 3)]
        g.set_sliders(rows); g.prepare()
        got = g.process_host(x, block=512)
        got_v = np.concatenate([ev.read_vars() for _, _, ev in g.shards], axis=0)
        st = g.reduce(shard_values=[1e-9, 3e-9])
    assert np.array_equal(got, want) and np.array_equal(got_v, want_v)
    assert st["n_shards"] == 2 and st["used_rccl"] == 1 and st["max_value"] == 3e-9 and st["units"] == n * 2 * frames


def test_bench_group_driver_on_two_devices():
    """`bench.py --gpus 2 --group`: the one-process multi-GPU driver end to end (null test on every shard included)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--group", "--steps", "2", "--warmup", "1", "--instances-total", "256",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["group"]["n_shards"] == 2 and line["group"]["used_rccl"] == 1
    assert line["null_test_dbfs"] <= -100.0


def test_group_errors_are_loud():
    import zabatch
    with pytest.raises(zabatch.ZabError):
        zabatch.Group("DDT", 1, devices=[0, 0])          # a shard would be empty
    with pytest.raises(zabatch.ZabError):
        zabatch.Group("DDT", 4, devices=[0, 99])         # no such device
