"""file_*() builtins over host-provided file slots against oracle/file_ref.py, the restatement of the
processor's runtime file handles (src/JSFXJuceProcessor.cpp:4893-5215), plus the "no MIDI ports" behaviour of midirecv /
midisend. No reference test pins these (parity unpinned); handle numbers, cursors and copied cells are compared exactly."""
from pathlib import Path

import numpy as np
import pytest

from conftest import assert_state_close

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
    # a script that never closes what it opens (TextureXY opens slot 0 in every @block): the reference's handle table just grows
    # (:4976-4981), so handle numbers keep counting up -- 4, 5, ... 15 -- and the newest one reads like any other
] + [(1, 2, 0, 0)] * 12 + [
    (4, 0, 0, 0), (6, 0, 0, 0), (7, 0, 400, 10),
    (3, 0, 0, 0),            # close handle 15 -> free list
    (2, 3, 0, 0),            # h1 = open(slot 3) reuses 15
    (6, 1, 0, 0),
    (1, 2, 0, 0),            # h0 -> new handle 16
    (4, 0, 0, 0),
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


def test_a_handle_that_left_the_window_is_refused_loudly():
    """The runtime keeps the state of the 8 most recent handles; the reference keeps all of them. Using an older one must not
    answer (differently from the reference) but raise the host-only error -- after 11 opens handle 1's cell holds handle 9."""
    from oracle import port
    if not port.port_path("fx_filekat").exists():
        pytest.skip("fixture port not built")
    p = port.Port("fx_filekat", 48000.0, mem_cap=MEM_CAP)
    it, ch, sr = _files()[2]
    p.file_slot_set(2, it, ch, sr)
    p.set_sliders([0, 0, 0, 0]); p.prepare()
    z = np.zeros((1, 4), np.float32)
    for k in range(11):
        p.set_sliders([1, 2, 0, 0]); p.process(z, 4)
        assert p.var("ret") == k + 1
    p.set_sliders([11, 0, 11, 0]); p.process(z, 4)           # the newest: fine
    assert p.var("ret") == 300 and p.err == 0
    p.set_sliders([11, 0, 4, 0]); p.process(z, 4)            # inside the window: fine
    assert p.var("ret") == 300 and p.err == 0
    p.set_sliders([11, 0, 1, 0]); p.process(z, 4)            # left the window
    assert p.err & 4


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


# ---- RIFF/WAVE ingestion (round 4) ---------------------------------------------------------------------------------
def _riff(fmt_tag, channels, rate, bits, payload, extensible=False, junk=True):
    """A RIFF/WAVE file image by hand: optional JUNK chunk of odd length (pad byte), fmt (plain or EXTENSIBLE), data."""
    import struct
    block = channels * bits // 8
    if extensible:
        sub = struct.pack("<H", fmt_tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, channels, rate, rate * block, block, bits, 22, bits, (1 << channels) - 1) + sub
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * block, block, bits)
    chunks = b""
    if junk:
        chunks += b"JUNK" + struct.pack("<I", 5) + b"\1\2\3\4\5" + b"\0"
    chunks += b"fmt " + struct.pack("<I", len(fmt)) + fmt
    chunks += b"data" + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_wav_reader_decodes_every_supported_format(tmp_path):
    """zab_wav_read (the C library's own RIFF/WAVE reader) against files built byte by byte here and against an independent
    decoder (scipy.io.wavfile) on files an independent writer produced: PCM 8 / 16 / 24 / 32, float 32 / 64, plain and
    WAVE_FORMAT_EXTENSIBLE headers, 1 to 3 channels, chunks of odd length. Host code only: runs without a GPU."""
    import zabatch
    from scipy.io import wavfile
    if not zabatch.runtime_path().exists():
        pytest.skip("libzabatch.so not built")
    rng = np.random.default_rng(7)
    frames = 37
    for ch in (1, 2, 3):
        x = rng.uniform(-1, 1, (frames, ch))
        i16 = np.round(x * 32767).astype("<i2")
        i32 = np.round(x * 2147483647).astype("<i4")
        i24 = (i32 >> 8).astype("<i4")
        u8 = (np.round(x * 127) + 128).astype(np.uint8)
        b24 = np.stack([(i24 >> s) & 255 for s in (0, 8, 16)], axis=-1).astype(np.uint8).tobytes()
        cases = [(1, 8, u8.tobytes(), (u8.astype(np.float32) - 128) / np.float32(128)),
                 (1, 16, i16.tobytes(), i16.astype(np.float32) / np.float32(32768)),
                 (1, 24, b24, i24.astype(np.float32) / np.float32(8388608)),
                 (1, 32, i32.tobytes(), (i32.astype(np.float64) / 2147483648.0).astype(np.float32)),
                 (3, 32, x.astype("<f4").tobytes(), x.astype(np.float32)),
                 (3, 64, x.astype("<f8").tobytes(), x.astype(np.float32))]
        for tag, bits, payload, want in cases:
            for ext in (False, True):
                p = tmp_path / f"t_{ch}_{tag}_{bits}_{int(ext)}.wav"
                p.write_bytes(_riff(tag, ch, 44100, bits, payload, extensible=ext))
                got, sr, gb, gf = zabatch.wav_read(p)
                assert (sr, gb, gf) == (44100, bits, tag == 3) and got.shape == (frames, ch)
                assert np.array_equal(got, want.reshape(frames, ch)), (ch, tag, bits, ext)
        # an independent writer and decoder
        q = tmp_path / f"s_{ch}.wav"
        wavfile.write(q, 22050, i16)
        got, sr, _, _ = zabatch.wav_read(q)
        sr2, ref = wavfile.read(q)
        assert sr == sr2 == 22050 and np.array_equal(got, (ref.reshape(frames, ch).astype(np.float32) / np.float32(32768)))
    with pytest.raises(zabatch.ZabError):
        zabatch.wav_read(tmp_path / "missing.wav")
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFF\x04\0\0\0WAVX")
    with pytest.raises(zabatch.ZabError):
        zabatch.wav_read(bad)
    adpcm = tmp_path / "adpcm.wav"
    adpcm.write_bytes(_riff(2, 1, 8000, 4, b"\0" * 16))
    with pytest.raises(zabatch.ZabError):
        zabatch.wav_read(adpcm)


IR_WAV = Path(__file__).resolve().parent / "fixtures" / "ir_100ms_stereo_pcm16.wav"


@pytest.mark.gpu
def test_psychoconvolver_with_an_impulse_response_loaded_from_a_wav_file():
    """The leaf the file slots exist for, fed the way a host would: a committed 0.1 s stereo PCM16 impulse response through
    zab_file_slot_load_wav, against the CPU port given the same file decoded by an independent reader (scipy)."""
    import zabatch
    from oracle import port
    from scipy.io import wavfile
    from zajit import noise
    if not zabatch.module_path("PsychoConvolver").exists() or not port.port_path("PsychoConvolver").exists():
        pytest.skip("PsychoConvolver not built")
    meta = zabatch.leaf_meta("PsychoConvolver")
    n, frames = 3, 12288
    x = noise.white_noise(range(n), frames)
    sr, pcm = wavfile.read(IR_WAV)
    items = (pcm.astype(np.float32) / np.float32(32768)).astype(np.float64).reshape(-1)
    with zabatch.Engine("PsychoConvolver", n, mem_cap=1 << 22) as e:
        info = e.file_slot_load_wav(0, IR_WAV)
        assert info == {"channels": 2, "sample_rate": 48000, "bits": 16, "is_float": 0, "frames": 4800}
        e.set_sliders(meta["default_sliders"]); e.prepare()
        y = e.process_host(x, block=512)
        v = e.read_vars(); names = e.var_names()
    assert sr == 48000
    for i in range(n):
        p = port.Port("PsychoConvolver", 48000.0, mem_cap=1 << 22)
        p.file_slot_set(0, items, 2, 48000.0)
        p.set_sliders(meta["default_sliders"]); p.prepare()
        want = p.process(x[i], 512)
        assert np.abs(y[i].astype(np.float64) - want).max() <= 1e-5
        assert_state_close(names, v[i], p.vars(), what=f"PsychoConvolver+wav vars[{i}]")
    assert np.abs(y).max() > 0.05          # (the convolver really ran)
