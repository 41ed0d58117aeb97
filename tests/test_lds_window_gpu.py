"""LDS window of the lane-per-instance process kernel (csrc/zart.h "LDS WINDOW", zabatch.hip choose_lmem): a launch that keeps
mem[0, K) of its instances in LDS must be indistinguishable -- audio, vars, arena, high-water marks, bit for bit -- from one
that reads the arena in HBM, across several zab_process calls (write-back on exit, reload on entry), host-side arena
writes between calls, and stores that land above the window."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEAVES = ["ERBTilt", "NeuroCV", "EasyExpander", "SpectralStabilizer", "fx_fuzz0", "fx_fuzz2", "fx_fuzz5"]


def _run(leaf, n, x, block, poke):
    import zabatch
    meta = zabatch.leaf_meta(leaf)
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    rows[:, 0] += np.linspace(0.0, 1.0, n) * 0.5            # instances differ a little
    # (the window belongs to the lane-per-instance kernel; leaves with a time-parallel kernel would take that one by default)
    with zabatch.Engine(leaf, n, mem_cap=1 << 14, max_block=block, path=zabatch.ZAB_PATH_GENERIC) as e:
        e.set_sliders(rows); e.prepare()
        third = x.shape[-1] // 3
        ys = [e.process_host(np.ascontiguousarray(x[..., :third]), block=block)]
        shape = e.launch_shape()
        if poke:                                            # host writes into the windowed part between launches
            e.write_mem(3, np.full((n, 2), 0.125))
        ys.append(e.process_host(np.ascontiguousarray(x[..., third:2 * third]), block=block))
        ys.append(e.process_host(np.ascontiguousarray(x[..., 2 * third:]), block=block))
        high = e.mem_high()
        top = int(high.max())
        return np.concatenate(ys, axis=-1), e.read_vars(), (e.read_mem(0, top) if top else None), high, shape


@pytest.mark.parametrize("poke", [False, True])
@pytest.mark.parametrize("leaf", LEAVES)
def test_window_is_invisible(leaf, poke, monkeypatch):
    import zabatch
    from zajit import noise
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    nch = int(zabatch.leaf_meta(leaf)["nch"])
    n, frames, block = 70, 1536, 256
    x = np.zeros((n, nch, frames), np.float32)
    x[:, :2] = noise.white_noise(range(n), frames)[:, :min(2, nch)]
    monkeypatch.delenv("ZAB_LMEM", raising=False)
    y1, v1, m1, h1, shape1 = _run(leaf, n, x, block, poke)
    monkeypatch.setenv("ZAB_LMEM", "0")
    y0, v0, m0, h0, shape0 = _run(leaf, n, x, block, poke)
    assert shape0[1] == 0
    if int(h1.max()) > 0:
        assert shape1[1] > 0, f"{leaf}: window expected (footprint {int(h1.max())} words), launch shape {shape1}"
    assert np.array_equal(h0, h1)
    assert np.array_equal(y0.view(np.uint32), y1.view(np.uint32))
    assert np.array_equal(v0.view(np.uint64), v1.view(np.uint64))
    if m0 is not None:
        assert np.array_equal(m0.view(np.uint64), m1.view(np.uint64))


def test_window_not_taken_when_it_would_serialise_the_batch():
    """A window is only worth having while every wavefront of the batch is resident at once; a batch too large for that
    keeps reading the arena."""
    import zabatch
    meta = zabatch.leaf_meta("SpectralStabilizer")
    n = 8192
    with zabatch.Engine("SpectralStabilizer", n, mem_cap=4096, max_block=64) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        e.process_host(np.zeros((n, 2, 64), np.float32), block=64)
        assert e.launch_shape()[1] == 0
