import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "zorakaudio-experimental-plugins_amd"
GOLDEN = ROOT / "tests" / "golden"
for p in (str(PKG), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

# Reference tolerances: src/JSFXCorrectnessCheck.h:34-35
AUDIO_EPS = 1.0e-5     # on float32-cast output samples
SCALAR_EPS = 1.0e-8    # sliders, vars[], mem[]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


def golden_input(g):
    from zajit import noise
    return noise.white_noise([int(g["seed_instance"])], int(g["frames"]), channels=int(g["nch"]))[0]


def dbfs(x: float) -> float:
    return float(20.0 * np.log10(max(x, 1e-300)))


def assert_state_close(names, got, want, eps=SCALAR_EPS, what="vars", skip=(), rel=False):
    """Reference comparator semantics (src/JSFXCorrectnessCheck.h:40-49): NaN==NaN, inf by equality, else abs <= eps.
    rel=True (comparisons between two of THIS repo's kernels, not against the reference VM): eps scales with the magnitude above
    1 -- a re-associated sum differs by an ulp, which on a variable of 3e11 (Texture's knee_t with its span at the floor) is 3e-5."""
    bad = []
    # EEL2 variable names are case-insensitive, the AOT compiler's are not (SURVEY 8 a-2 addendum): a script that uses both `PI`
    # and `pi` (Spectral/Texture) has ONE variable in the reference VM and two in the compiled path, so those names cannot be
    # compared with a VM fixture; everything else of such a leaf can.
    lower = {}
    for n in names:
        lower.setdefault(str(n).lower(), []).append(str(n))
    skip = set(skip) | {n for group in lower.values() if len(group) > 1 for n in group}
    for i, n in enumerate(names):
        n = str(n)
        if n in skip:
            continue
        a, b = float(got[i]), float(want[i])
        if np.isnan(b):            # the EEL2 VM never created this variable (case-insensitive alias etc.)
            continue
        if np.isnan(a) or np.isinf(a) or np.isinf(b):
            ok = (np.isnan(a) and np.isnan(b)) or a == b
        else:
            ok = abs(a - b) <= eps * (max(1.0, abs(a), abs(b)) if rel else 1.0)
        if not ok:
            bad.append((n, a, b))
    assert not bad, f"{what} mismatch (first 8 of {len(bad)}): {bad[:8]}"


@pytest.fixture(scope="session")
def have_gpu():
    import torch
    return torch.cuda.is_available()


def leaf_of(case: str) -> str:
    """'DDT_far_extreme' -> 'DDT', 'fx_stft_default' -> 'fx_stft' (fixture leaves carry the fx_ prefix)."""
    parts = case.split("_")
    return "_".join(parts[:2]) if parts[0] == "fx" else parts[0]
