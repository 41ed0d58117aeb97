"""Host-buffer pipeline of zab_process (ZAB_BUF_HOST): long buffers are cut into time chunks of whole host blocks whose
copy-in, kernels and copy-out overlap on three HIP streams. The result must be what the unpipelined call gives, bit for bit
(the chunks are ordinary consecutive launches): hand-written DDT kernel, generic kernels, a Faust leaf, a message-bus leaf;
pageable and page-locked (zab_host_alloc) buffers; a ragged tail; a padded frame stride."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(leaf, x, block, chunk_kb, monkeypatch, pinned=False, n_calls=1):
    import zabatch
    monkeypatch.setenv("ZAB_PIPE_CHUNK_KB", str(chunk_kb))
    meta = zabatch.leaf_meta(leaf)
    n = x.shape[0]
    with zabatch.Engine(leaf, n, max_block=block) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        ys = []
        for part in np.array_split(x, n_calls, axis=-1):
            part = np.ascontiguousarray(part)
            if pinned:
                with zabatch.PinnedArray(part.shape) as pi, zabatch.PinnedArray(part.shape) as po:
                    pi.array[...] = part
                    e.process_host(pi.array, block=block, out=po.array)
                    ys.append(po.array.copy())
            else:
                ys.append(e.process_host(part, block=block))
        launches = e.last_timing()[1]
        return np.concatenate(ys, axis=-1), e.read_vars(), launches


@pytest.mark.parametrize("leaf,n,frames,block", [("DDT", 24, 20000, 512), ("ERBTilt", 70, 9000, 256), ("ClickBeGoneSG", 33, 12000, 512),
                                                 ("IPCProbeA", 3, 6000, 64), ("NeuroCV", 5, 7000, 512)])
def test_pipelined_equals_whole(leaf, n, frames, block, monkeypatch):
    import zabatch
    from zajit import noise
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    nch = int(zabatch.leaf_meta(leaf)["nch"])
    x = np.zeros((n, nch, frames), np.float32)
    x[:, :2] = noise.white_noise(range(n), frames)[:, :min(2, nch)]
    y0, v0, l0 = _run(leaf, x, block, 0, monkeypatch)                       # pipeline off
    y1, v1, l1 = _run(leaf, x, block, 16, monkeypatch)                      # 16 KB chunks: many of them, ragged tail
    y2, v2, l2 = _run(leaf, x, block, 16, monkeypatch, pinned=True)
    assert l1 > l0 or leaf == "IPCProbeA", (l0, l1)                         # it did go through the chunked path (the bus
                                                                            # leaf launches per host block either way)
    for y, v in ((y1, v1), (y2, v2)):
        if leaf == "DDT":      # hand-written kernel: meter / smoother state is a reduction over the launch, so its rounding
            assert np.abs(y0.astype(np.float64) - y).max() <= 1e-5 and np.abs(v0 - v).max() <= 1e-8     # depends on the split
            continue
        assert np.array_equal(y0.view(np.uint32), y.view(np.uint32))
        assert np.array_equal(v0.view(np.uint64), v.view(np.uint64))


def test_pipelined_with_padded_stride(monkeypatch):
    """frame_stride > frames on the host side: rows are copied with their pitch."""
    import zabatch
    from zajit import noise
    monkeypatch.setenv("ZAB_PIPE_CHUNK_KB", "8")
    n, frames, stride, block = 6, 8192, 8192 + 96, 512
    meta = zabatch.leaf_meta("DDT")
    x = noise.white_noise(range(n), frames)
    xin = np.full((n, 2, stride), 7.0, np.float32); xin[:, :, :frames] = x
    out = np.full((n, 2, stride), -3.0, np.float32)
    with zabatch.Engine("DDT", n) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        e._chk(e.L.zab_process(e.h, xin.ctypes.data, out.ctypes.data, frames, stride, block, zabatch.ZAB_BUF_HOST))
    with zabatch.Engine("DDT", n) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        monkeypatch.setenv("ZAB_PIPE_CHUNK_KB", "0")
        ref = e.process_host(x, block=block)
    assert np.abs(out[:, :, :frames].astype(np.float64) - ref).max() <= 1e-5
    assert np.all(out[:, :, frames:] == -3.0)                               # the padding is left alone
