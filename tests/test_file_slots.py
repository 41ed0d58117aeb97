"""file_*() builtins over host-provided file slots against oracle/file_ref.py, the restatement of the
processor's runtime file handles (src/JSFXJuceProcessor.cpp:4893-5215), plus the "no MIDI ports" behaviour of midirecv /
midisend. No reference test pins these (parity unpinned); handle numbers, cursors and copied cells are compared exactly."""
import numpy as np
import pytest

MEM_CAP = 65536
# (op, a, b, c) scripts; a selects handle slot h0 / h1 for ops >= 3
SCRIPT = [
    (1, 5, 0, 0),            # open an unassigned slot -> -1
    (1, 2, 0, 0),            # h0 = open(slot 2) -> 1
    (2, 3, 0, 0),            # h1 = open(slot 3) -> 2
    (4, 0, 0, 0), (4, 1, 0, 0),          # avail
    (5, 0, 0, 0), (5, 1, 0, 0),          # riff
    (6, 0, 0, 0), (6, 0, 0, 0),          # var x2 advances
    (7, 0, 100, 50), (7, 0, 200.99999, 1000), (7, 0, 300, 5),   # mem: partial, to the end, nothing left
    (4, 0, 0, 0),
    (8, 0, 0, 0), (4, 0, 0, 0),          # rewind
    (9, 1, 7.99999, 0), (6, 1, 0, 0), (9, 1, 1e9, 0), (6, 1, 0, 0), (9, 1, -3, 0),   # seek clamps
    (3, 0, 0, 0),            # close h0 -> handle 1 goes to the free list
    (4, 0, 0, 0),            # avail on the closed handle -> 0
    (2, 2, 0, 0),            # h1 = open(slot 2) reuses handle 1 (LIFO)
    (1, 3, 0, 0),            # h0 = open(slot 3) -> new handle 3
    (11, 0, 2, 0), (11, 0, 9, 0), (11, 0, 0, 0), (11, 0, -1, 0),   # raw handle numbers
    (10, 0, 0, 0),
    (7, 1, 65530, 20),       # would run past the arena: documented deviation (skipped below)
]


def _files():
    rng = np.random.default_rng(5)
    return {2: (rng.standard_normal(300), 2, 44100.0), 3: (rng.standard_normal(17), 1, 48000.0)}


def _run(make):
    from oracle import file_ref
    files = _files()
    ref = file_ref.FileRef()
    for k, (it, ch, sr) in files.items():
        ref.assign(k, it, ch, sr)
    mem = np.zeros(MEM_CAP)
    hv = {"h0": 0.0, "h1": 0.0, "nch": -7.0, "sr": -7.0, "v": -7.0}
    dut = make(files)
    for step, (op, a, b, c) in enumerate(SCRIPT):
        if op == 7 and b + c > MEM_CAP:
            continue                      # the reference grows mem; the fixed arena reports overflow instead
        h = hv["h1"] if a else hv["h0"]
        if op == 1: want = hv["h0"] = ref.open(a)
        elif op == 2: want = hv["h1"] = ref.open(a)
        elif op == 3: want = ref.close(h)
        elif op == 4: want = ref.avail(h)
        elif op == 5: want, hv["nch"], hv["sr"] = ref.riff(h, hv["nch"], hv["sr"])
        elif op == 6: want, hv["v"] = ref.var(h)
        elif op == 7: want = ref.mem(mem, h, b, c)
        elif op == 8: want = ref.rewind(h)
        elif op == 9: want = ref.seek(h, b)
        elif op == 10: want = 0.0
        elif op == 11: want = ref.avail(b)
        got = dut(op, a, b, c)
        tag = (step, op, a, b, c)
        assert got["ret"] == want, (tag, got, want)
        for k in hv:
            assert got[k] == hv[k], (tag, k, got[k], hv[k])
        assert np.array_equal(got["mem"], mem), tag


def test_port_file_slots():
    from oracle import port
    if not port.port_path("fx_filekat").exists():
        pytest.skip("fixture port not built")

    def make(files):
        p = port.Port("fx_filekat", 48000.0, mem_cap=MEM_CAP)
        for k, (it, ch, sr) in files.items():
            p.file_slot_set(k, it, ch, sr)
        p.set_sliders([0, 0, 0, 0]); p.prepare()

        def step(op, a, b, c):
            p.set_sliders([op, a, b, c])
            p.process(np.zeros((1, 4), np.float32), 4)
            out = {k: p.var(k) for k in ("ret", "h0", "h1", "nch", "sr", "v")}
            out["mem"] = p.mem(0, MEM_CAP)
            return out
        return step
    _run(make)


def test_port_midi_without_ports():
    """No MIDI queues: midirecv returns 0 and leaves its outputs alone, midisend returns 0 (dropped)."""
    from oracle import port
    if not port.port_path("fx_filekat").exists():
        pytest.skip("fixture port not built")
    p = port.Port("fx_filekat", 48000.0, mem_cap=MEM_CAP)
    p.set_sliders([0, 0, 0, 0]); p.prepare()
    p.set_sliders([12, 0, 0, 0]); p.process(np.zeros((1, 4), np.float32), 4)
    assert p.var("ret") == 0.0 and p.var("m1") == 0.0


@pytest.mark.gpu
def test_gpu_file_slots():
    import zabatch

    def make(files):
        e = zabatch.Engine("fx_filekat", 3, mem_cap=MEM_CAP)
        for k, (it, ch, sr) in files.items():
            e.file_slot_set(k, it, ch, sr)
        e.set_sliders([0, 0, 0, 0]); e.prepare()
        names = e.var_names()

        def step(op, a, b, c):
            e.set_sliders([op, a, b, c])
            e.process_host(np.zeros((3, 1, 4), np.float32), block=4)
            v = e.read_vars()
            assert np.array_equal(v[0], v[2]), "instances share the slots but own their handles: same script, same state"
            out = {k: v[1][names.index(k)] for k in ("ret", "h0", "h1", "nch", "sr", "v")}
            out["mem"] = e.read_mem(0, MEM_CAP)[1]
            return out
        return step
    _run(make)
