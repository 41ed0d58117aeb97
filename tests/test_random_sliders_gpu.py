"""Every JSFX leaf at RANDOM slider settings, device vs CPU port: the VM fixtures pin the default settings; this walks other
branches of the scripts (mode switches, bypasses, extreme times and gains). Slider rows are drawn inside each slider's declared
range and snapped to its step like a host parameter would be (src/JSFXJuceProcessor.cpp:5556-5596), one row per instance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEAVES = ["ADS", "ATTACK", "Alias", "BedRock", "DPT", "DDT", "DOT", "ERBTilt", "EasyExpander", "NeuroCV", "RTT", "Roomalizer",
          "SOMA", "SaliencePush", "SpectralStabilizer", "TSEQ", "PsychoConvolver"]      # (CMD couples its instances through gmem)
CAPS = {"Alias": 1 << 19, "SOMA": 1 << 18, "PsychoConvolver": 1 << 22}


def _rows(meta, n, seed):
    rng = np.random.default_rng(seed)
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    for k, sp in meta["sliders"].items():
        k = int(k)
        lo, hi, step = float(sp["min"]), float(sp["max"]), float(sp["step"])
        if sp.get("is_string") or not np.isfinite(lo) or not np.isfinite(hi) or hi <= lo:
            continue
        v = lo + rng.random(n) * (hi - lo)
        if step > 0:
            v = lo + np.round((v - lo) / step) * step
        rows[:, k] = np.clip(v, lo, hi)
    rows[0] = np.array(meta["default_sliders"], dtype=np.float64)        # instance 0 stays at the defaults
    return rows


@pytest.mark.parametrize("leaf", LEAVES)
def test_random_slider_rows_device_vs_port(leaf):
    import zabatch
    from oracle import port
    from zajit import noise
    from conftest import AUDIO_EPS, assert_state_close
    if not zabatch.module_path(leaf).exists() or not port.port_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    nch, n, frames, block = int(meta["nch"]), 6, 3072, 512
    cap = CAPS.get(leaf, 1 << 16)
    rows = _rows(meta, n, seed=sum(map(ord, leaf)) + 1000 * int(os.environ.get("ZA_RAND_SEED", "0")))     # (other draws: ZA_RAND_SEED=k)
    x = np.zeros((n, nch, frames), np.float32)
    x[:, :2] = noise.white_noise(range(n), frames)[:, :min(2, nch)]
    with zabatch.Engine(leaf, n, mem_cap=cap, max_block=block) as e:
        e.set_sliders(rows); e.prepare()
        y = e.process_host(x, block=block)
        v = e.read_vars(); names = e.var_names()
    for i in range(n):
        p = port.Port(leaf, 48000.0, mem_cap=cap)
        p.set_sliders(rows[i]); p.prepare()
        ref = p.process(x[i], block)
        err = np.abs(y[i].astype(np.float64) - ref).max()
        assert err <= AUDIO_EPS, (leaf, i, err, rows[i][:12])
        assert_state_close(names, v[i], p.vars(), what=f"{leaf} vars[{i}] sliders {rows[i][:8]}")
