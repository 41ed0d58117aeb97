"""The state object of a section must not be pinned in scratch memory by the runtime's out-of-line builtins (csrc/zart.h ZaEnv /
ZA_OUTCALL): read the built code objects and check the process kernels of leaves that call such
builtins from their audio path. Before the split fx_ringio's process kernel held 1448 B of scratch and 91 scratch
instructions; what remains is the environment copy around the call (~300 B)."""
import sys
from pathlib import Path
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

# (fx_stft: an FFT leaf's kernels are capped at 256 registers -- two wavefronts per SIMD, zab_generic.hip.h ZA_OCC -- and its
#  process kernel spills ~0.9 KB of temporaries for that; the state object itself is in registers, the bar is still below the
#  1448 B of the pinned form)
CASES = [("fx_ringio", 512), ("fx_stft", 1100), ("fx_fftbench", 512), ("fx_msgkat", 512), ("fx_gmemkat", 512),
         ("fx_filekat", 512), ("fx_poolkat", 512), ("fx_randkat", 64)]


@pytest.mark.parametrize("leaf,limit", CASES)
def test_process_kernel_keeps_the_state_in_registers(leaf, limit):
    import kernel_resources as kr
    if not (kr.LLVM / "llvm-readelf").exists():
        pytest.skip("no ROCm llvm tools")
    so = kr.build.LIB / f"libzab_{leaf}.so"
    assert so.exists(), f"{so} not built (python -c 'import __graft_entry__ as g; g.build()')"
    res = {k: (scr, vg) for k, scr, vg in kr.kernel_resources(so)}
    name = f"zab_{leaf}_process"
    assert name in res, sorted(res)
    assert res[name][0] <= limit, f"{name}: {res[name][0]} B of scratch per lane -- is a runtime builtin taking the state object by reference again?"
