"""FFT builtins against known answers produced by the reference's own WDL build
(tests/golden/wdl_fft.npz: WDL_fft / WDL_real_fft / WDL_fft_permute at REALSIZE=8).

CPU: through the port of tests/fixtures/fftkat.jsfx (zart_fft.h compiled by g++).
GPU: the same fixture leaf through the C ABI (zart_fft.h as __device__ code), plus the STFT fixture leaf against the
reference VM's golden run.
"""
import numpy as np
import pytest

from conftest import AUDIO_EPS, GOLDEN, SCALAR_EPS, assert_state_close, golden_input, load_golden

KAT = np.load(GOLDEN / "wdl_fft.npz")
SIZES = (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768)     # every size the builtins accept
OPS = {"fft": 1, "ifft": 2, "fft_real": 3, "ifft_real": 4, "fft_permute": 5, "fft_ipermute": 6, "convolve_c": 7}
SRC_BASE = 65536          # convolve_c's second operand: the next 65536-cell page (a region may not cross a page)


def _wdl_perm(n):
    """WDL_fft_permute_tab(n) from the recursion zart_fft.h uses (pinned against the reference's tables for n <= 4096 below)."""
    def freq(i, m_):
        mul, add, mask, nn = 1, 0, m_ - 1, m_
        while nn > 2:
            m = nn >> 1
            if i < m:
                mul <<= 1; nn = m; continue
            i -= m; m >>= 1
            if i < m:
                add += mul; mul <<= 2; nn = m; continue
            i -= m
            add -= mul; mul <<= 2; nn = m
        return (i * mul + add) & mask
    perm = np.zeros(n, dtype=np.int64)
    for i in range(n):
        perm[(n - freq(i, n)) & (n - 1)] = i
    return perm


def _pack(c):
    out = np.empty(2 * len(c)); out[0::2] = c.real; out[1::2] = c.imag
    return out


def _numpy_vectors(n, z, r):
    """WDL's conventions (src/WDL/fft.h:55-73) over an INDEPENDENT transform (numpy's): forward = the DFT scattered into WDL's
    output order; inverse = its input taken in that order, unscaled; the real pair = n / 2 packed bins in the order of n / 2,
    scaled by 2, DC / Nyquist in bin 0. Equal to the reference library's own output to rounding for every size
    (test_numpy_construction_equals_the_reference_library, dev container), so the GPU box needs no reference-built binary."""
    perm, permh, h = _wdl_perm(n), _wdl_perm(n // 2), n // 2
    zc = z[0::2] + 1j * z[1::2]
    fwd = np.empty(n, complex); fwd[perm] = np.fft.fft(zc)
    inv = np.fft.ifft(zc[perm]) * n
    R = np.fft.rfft(r) * 2
    nb = np.empty(h, complex); nb[0] = R[0].real + 1j * R[h].real; nb[1:] = R[1:h]
    packed = np.empty(h, complex); packed[permh] = nb
    nbin = (r[0::2] + 1j * r[1::2])[permh]
    full = np.empty(h + 1, complex); full[0] = nbin[0].real; full[h] = nbin[0].imag; full[1:h] = nbin[1:]
    return {"c_in": z, "c_fwd": _pack(fwd), "c_inv": _pack(inv), "r_in": r, "r_fwd": _pack(packed),
            "r_inv": np.fft.irfft(full, n) * n, "perm": perm}


def _vectors(n):
    """Known answers: the reference's WDL build for n <= 4096 (committed, wdl_fft.npz); above that -- the vectors would be
    megabytes of incompressible doubles -- seeded inputs through _numpy_vectors."""
    if f"c{n}_in" in KAT:
        return {k: KAT[f"{k[0]}{n}{k[1:]}"] for k in ("c_in", "c_fwd", "c_inv", "r_in", "r_fwd", "r_inv")} | {"perm": KAT[f"perm{n}"]}
    rng = np.random.default_rng(20261004 + n)
    return _numpy_vectors(n, rng.standard_normal(2 * n), rng.standard_normal(n))


@pytest.mark.parametrize("n", SIZES)
def test_numpy_construction_equals_the_reference_library(n):
    """Pins _numpy_vectors: against the committed WDL vectors up to 4096 points (anywhere), against the live reference library
    (oracle/_ref, dev container) above."""
    tol = lambda want: 64 * n * np.finfo(np.float64).eps * max(1.0, np.abs(want).max())
    if f"c{n}_in" in KAT:
        ref = {k: KAT[f"{k[0]}{n}{k[1:]}"] for k in ("c_in", "c_fwd", "c_inv", "r_in", "r_fwd", "r_inv")} | {"perm": KAT[f"perm{n}"]}
    else:
        from oracle import eel_oracle
        if not eel_oracle.available():
            pytest.skip("oracle/_ref not built")
        rng = np.random.default_rng(20261004 + n)
        z, r = rng.standard_normal(2 * n), rng.standard_normal(n)
        ref = {"c_in": z, "c_fwd": eel_oracle.wdl_fft(z, n, False), "c_inv": eel_oracle.wdl_fft(z, n, True), "r_in": r,
               "r_fwd": eel_oracle.wdl_real_fft(r, n, False), "r_inv": eel_oracle.wdl_real_fft(r, n, True),
               "perm": eel_oracle.wdl_fft_permute(n)}
    got = _numpy_vectors(n, ref["c_in"], ref["r_in"])
    assert np.array_equal(got["perm"], ref["perm"])
    for k in ("c_fwd", "c_inv", "r_fwd", "r_inv"):
        assert np.abs(got[k] - ref[k]).max() <= tol(ref[k]), k


def _cases(n):
    """(op, input doubles, expected doubles, second operand or None)"""
    V = _vectors(n)
    perm = V["perm"]
    cin, cf, ci = V["c_in"], V["c_fwd"], V["c_inv"]
    z = cin[0::2] + 1j * cin[1::2]
    nat = np.empty(2 * n); nat[0::2] = z[perm].real; nat[1::2] = z[perm].imag          # natural[k] = buf[perm[k]]
    wdl = np.empty_like(z); wdl[perm] = z
    ip = np.empty(2 * n); ip[0::2] = wdl.real; ip[1::2] = wdl.imag                      # buf[perm[k]] = natural[k]
    yield "fft", cin, cf, None
    yield "ifft", cin, ci, None
    yield "fft_permute", cin, nat, None
    yield "fft_ipermute", cin, ip, None
    yield "fft_real", V["r_in"], V["r_fwd"], None
    yield "ifft_real", V["r_in"], V["r_inv"], None
    # convolve_c: dest[i] *= src[i] over n complex pairs, the reference's own four products and two sums
    # (src/JSFXJuceProcessor.cpp:1370-1380; no fused multiply-add on either side) -> bit-exact
    src = np.roll(cin, 7) * 0.5
    ar, ai, br, bi = cin[0::2], cin[1::2], src[0::2], src[1::2]
    want = np.empty(2 * n); want[0::2] = ar * br - ai * bi; want[1::2] = ar * bi + ai * br
    yield "convolve_c", cin, want, src


def _tol(n, want):
    return 64 * n * np.finfo(np.float64).eps * max(1.0, np.abs(want).max())


def test_permutation_recursion_matches_wdl_table():
    """za_fft_freq restated in numpy == WDL_fft_permute for every size in the fixture."""
    def freq(i, n):
        mul, add, mask = 1, 0, n - 1
        while n > 2:
            m = n >> 1
            if i < m:
                mul <<= 1; n = m; continue
            i -= m; m >>= 1
            if i < m:
                add += mul; mul <<= 2; n = m; continue
            i -= m
            add -= mul; mul <<= 2; n = m
        return (i * mul + add) & mask
    for n in (16, 32, 128, 2048, 4096):
        perm = np.zeros(n, dtype=np.int64)
        for i in range(n):
            perm[(n - freq(i, n)) & (n - 1)] = i
        assert np.array_equal(perm, KAT[f"perm{n}"]), n


@pytest.mark.parametrize("n", SIZES)
def test_port_fft_known_answers(n):
    from oracle import port
    if not port.port_path("fx_fftkat").exists():
        pytest.skip("fixture port not built")
    for op, x, want, src in _cases(n):
        p = port.Port("fx_fftkat", 48000.0, mem_cap=1 << 18)
        p.set_sliders([0, n, 0, SRC_BASE]); p.prepare()
        p.mem_write(0, x)
        if src is not None:
            p.mem_write(SRC_BASE, src)
        p.set_sliders([OPS[op], n, 0, SRC_BASE]); p.run_slider()
        got = p.mem(0, len(want))
        assert p.err == 0
        tol = 0.0 if op == "convolve_c" else _tol(n, want)
        assert np.abs(got - want).max() <= tol, (op, n, np.abs(got - want).max())


def test_port_fft_argument_rules():
    """Page-crossing regions and bad sizes are silent no-ops (src/JSFXJuceProcessor.cpp:1126-1195); convolve_c; memcpy."""
    from oracle import port
    if not port.port_path("fx_fftkat").exists():
        pytest.skip("fixture port not built")
    rng = np.random.default_rng(5)
    x = rng.standard_normal(64)
    for n, base in ((24, 0), (8, 0), (65536, 0), (32, 65536 - 32)):      # not pow2, too small, too big, crosses a page
        p = port.Port("fx_fftkat", 48000.0, mem_cap=1 << 18)
        p.set_sliders([0, n, base, 0]); p.prepare()
        p.mem_write(base, x)
        p.set_sliders([1, n, base, 0]); p.run_slider()
        assert np.array_equal(p.mem(base, 64), x), (n, base)
    # convolve_c: dest *= src, 16 pairs, disjoint and overlapping (src read before dest is written)
    a, b = rng.standard_normal(32), rng.standard_normal(32)
    za, zb = a[0::2] + 1j * a[1::2], b[0::2] + 1j * b[1::2]
    p = port.Port("fx_fftkat", 48000.0, mem_cap=1 << 17)
    p.set_sliders([0, 16, 0, 100]); p.prepare()
    p.mem_write(0, a); p.mem_write(100, b)
    p.set_sliders([7, 16, 0, 100]); p.run_slider()
    got = p.mem(0, 32)
    assert np.allclose(got[0::2] + 1j * got[1::2], za * zb, rtol=0, atol=1e-14)
    p = port.Port("fx_fftkat", 48000.0, mem_cap=1 << 17)
    both = rng.standard_normal(40)
    p.set_sliders([0, 16, 8, 0]); p.prepare()
    p.mem_write(0, both)
    p.set_sliders([7, 16, 8, 0]); p.run_slider()                           # dest = [8,40), src = [0,32): overlap
    got = p.mem(8, 32)
    zd, zs = both[8:40][0::2] + 1j * both[8:40][1::2], both[0:32][0::2] + 1j * both[0:32][1::2]
    assert np.allclose(got[0::2] + 1j * got[1::2], zd * zs, rtol=0, atol=1e-14)


def test_port_stft_fixture_matches_reference_vm():
    from oracle import port
    if not port.port_path("fx_stft").exists():
        pytest.skip("fixture port not built")
    g = load_golden("fx_stft_default")
    p = port.Port("fx_stft", float(g["srate"]), mem_cap=1 << 16)
    p.set_sliders(g["sliders"]); p.prepare()
    y = p.process(golden_input(g), int(g["block"]))
    assert np.abs(y.astype(np.float64) - g["out"]).max() <= AUDIO_EPS
    names = [str(s) for s in g["var_names"]]
    assert_state_close(names, p.vars(), g["vars"], what="stft vars")
    want = np.zeros(int(g["mem_high"])); want[g["mem_idx"]] = g["mem_val"]
    assert np.abs(p.mem(0, len(want)) - want).max() <= SCALAR_EPS


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", ["fx_fftkat", "fx_fftkat_full"])   # default: 1024-point LDS buffer, larger complex transforms sliced; _full: all of it in LDS
@pytest.mark.parametrize("n", SIZES)
def test_gpu_fft_known_answers(n, leaf):
    """Every op on 70 instances at once (two waves), each instance with its own scaled copy of the input."""
    import zabatch
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    inst = 70
    scale = 1.0 + np.arange(inst)[:, None] * 0.125
    if n > 4096:
        inst = 6                                    # (some ops of these sizes take the serial device transform: keep the batch small)
        scale = 1.0 + np.arange(inst)[:, None] * 0.125
    for op, x, want, src in _cases(n):
        with zabatch.Engine(leaf, inst, mem_cap=1 << 18) as e:
            e.set_sliders([0, n, 0, SRC_BASE]); e.prepare()
            e.write_mem(0, scale * x[None, :])
            if src is not None:
                e.write_mem(SRC_BASE, np.repeat(src[None, :], inst, axis=0))
            e.set_sliders([OPS[op], n, 0, SRC_BASE])
            e.process_host(np.zeros((inst, 2, 8), np.float32), block=8)
            got = e.read_mem(0, len(want))
        ref = scale * want[None, :]
        tol = _tol(n, ref)
        if op == "convolve_c":                      # bit-exact: the reference's four products and two sums on each instance's own data
            d = scale * x[None, :]
            ar, ai, br, bi = d[:, 0::2], d[:, 1::2], src[None, 0::2], src[None, 1::2]
            ref = np.empty_like(d); ref[:, 0::2] = ar * br - ai * bi; ref[:, 1::2] = ar * bi + ai * br
            tol = 0.0
        assert np.abs(got - ref).max() <= tol, (op, n, np.abs(got - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["auto", "generic"])
@pytest.mark.parametrize("mask", [1, 9, 31])       # fft alone (WDL order out); fft, ifft; the four builtins as fused natural-order pairs
@pytest.mark.parametrize("size", [2048, 4096])
def test_gpu_sliced_transforms_keep_the_bits_of_the_in_lds_form(size, mask, path):
    """Transforms beyond the 1024-point LDS buffer run in two stages of 1024-point chunks (zart_fft.h, za_fft_coop); the *_full
    build keeps the whole transform in LDS. Same butterflies, twiddles and order per element: the arenas must be identical,
    on full and on partly filled wavefronts (70 instances) and on both kernels."""
    import zabatch
    if not zabatch.module_path("fx_fftbench_full").exists():
        pytest.skip("fx_fftbench_full not built")
    p = {"auto": zabatch.ZAB_PATH_AUTO, "generic": zabatch.ZAB_PATH_GENERIC}[path]
    got = {}
    for leaf in ("fx_fftbench", "fx_fftbench_full"):
        with zabatch.Engine(leaf, 70, mem_cap=1 << 17, path=p) as e:
            row = np.zeros(64); row[0] = size; row[1] = 1; row[2] = mask
            e.set_sliders(row); e.prepare()
            e.process_host(np.zeros((70, e.nch, 64), np.float32), block=64)
            got[leaf] = e.read_mem(0, 2 * size)
    assert np.isfinite(got["fx_fftbench"]).all() and np.abs(got["fx_fftbench"]).max() > 1.0
    assert np.array_equal(got["fx_fftbench"], got["fx_fftbench_full"])
    assert np.array_equal(got["fx_fftbench"][0], got["fx_fftbench"][69])


@pytest.mark.gpu
def test_gpu_stft_fixture_matches_reference_vm():
    import zabatch
    g = load_golden("fx_stft_default")
    n = 66
    x = np.repeat(golden_input(g)[None], n, axis=0)
    with zabatch.Engine("fx_stft", n, srate=float(g["srate"])) as e:
        e.set_sliders(g["sliders"]); e.prepare()
        y = e.process_host(x, block=int(g["block"]))
        v = e.read_vars()
        mem = e.read_mem(0, int(g["mem_high"]))
        names = e.var_names()
    assert np.abs(y.astype(np.float64) - g["out"].astype(np.float64)[None]).max() <= AUDIO_EPS
    want = np.zeros(int(g["mem_high"])); want[g["mem_idx"]] = g["mem_val"]
    for i in (0, 63, 65):
        assert_state_close(names, v[i], g["vars"], what=f"stft vars[{i}]")
    assert np.abs(mem - want[None]).max() <= SCALAR_EPS


@pytest.mark.gpu
@pytest.mark.parametrize("path,n", [("auto", 256), ("generic", 24)])
def test_config_c3_shape_256_instances_of_eight_channels(path, n):
    """BASELINE config C3 as it is written -- 4096-point STFT, hop 1024, 256 instances x 8 channels -- against the reference VM's
    run of the eight-channel fixture (tests/fixtures/stft4k8.jsfx): every instance gets the fixture's input; sampled instances
    are compared in full (audio, vars, arena, high-water mark), all of them on audio."""
    import zabatch
    if not zabatch.module_path("fx_stft4k8").exists():
        pytest.skip("fx_stft4k8 not built")
    g = load_golden("fx_stft4k8_default")
    x = np.repeat(golden_input(g)[None], n, axis=0)
    assert x.shape[1] == 8
    p = {"auto": zabatch.ZAB_PATH_AUTO, "generic": zabatch.ZAB_PATH_GENERIC}[path]
    with zabatch.Engine("fx_stft4k8", n, srate=float(g["srate"]), mem_cap=1 << 17, path=p) as e:
        assert e.nch == 8
        e.set_sliders(g["sliders"]); e.prepare()
        y = e.process_host(x, block=int(g["block"]))
        assert e.used_fast_path() == (path == "auto")
        v = e.read_vars()
        names = e.var_names()
        high = e.mem_high()
        picks = sorted({0, n // 3, n - 1})
        mems = {i: e.read_mem(0, int(g["mem_high"]), first=i, count=1)[0] for i in picks}
    assert np.abs(y.astype(np.float64) - g["out"].astype(np.float64)[None]).max() <= AUDIO_EPS
    want = np.zeros(int(g["mem_high"])); want[g["mem_idx"]] = g["mem_val"]
    for i in picks:
        assert_state_close(names, v[i], g["vars"], what=f"stft4k8 vars[{i}]")
        assert np.abs(mems[i] - want).max() <= SCALAR_EPS
    assert (high == int(g["mem_high"])).all()
