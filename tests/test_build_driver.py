"""zajit.build as the drop-in build driver (scripts/build.py of the reference): --only takes what the reference takes
(category, key, slug, name, path, bundleId, clapId: scripts/pluginlib.py:243-257), --correctness-check runs a built leaf on the
device against the reference shadow VM's recorded runs (zajit/check.py) and fails the build above the reference's tolerances."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import PKG, ROOT

REF = Path("/root/reference/plugins")


def test_only_matches_what_the_reference_build_matches():
    from zajit import build
    leaves = {"DDT": {"category": "Spatialization", "dir": "plugins/Spatialization/DDT",
                      "meta": {"name": "DDT", "slug": "DDT", "bundleId": "com.zorakaudio.experimental.ddt", "clapId": "com.zorakaudio.experimental.ddt"}},
              "DOT": {"category": "Spatialization", "dir": "plugins/Spatialization/DOT",
                      "meta": {"name": "Delay-Oriented Thing", "slug": "DOT", "bundleId": "com.zorakaudio.experimental.dot", "clapId": "x.dot"}},
              "EasyExpander": {"category": "Dynamics", "dir": "plugins/Dynamics/EasyExpander",
                               "meta": {"name": "EasyExpander", "slug": "EasyExpander", "bundleId": "com.zorakaudio.experimental.easyexpander", "clapId": "c"}}}
    assert build.select(leaves, []) == ["DDT", "DOT", "EasyExpander"]
    assert build.select(leaves, ["ddt"]) == ["DDT"]                                # a key, exactly
    assert build.select(leaves, ["Spatialization"]) == ["DDT", "DOT"]              # a category
    assert build.select(leaves, ["spatialization/dot"]) == ["DOT"]                 # a path
    assert build.select(leaves, ["delay-oriented"]) == ["DOT"]                     # a name
    assert build.select(leaves, ["experimental.easyexpander"]) == ["EasyExpander"]  # a bundleId
    assert build.select(leaves, ["x.dot", "dynamics"]) == ["DOT", "EasyExpander"]  # a clapId, then a category
    assert build.select(leaves, ["nothing-like-this"]) == []


def test_cli_knows_the_reference_flags():
    out = subprocess.run([sys.executable, "-m", "zajit.build", "--help"], cwd=PKG, capture_output=True, text=True).stdout
    for flag in ("--only", "--list", "--correctness-check"):
        assert flag in out
    r = subprocess.run([sys.executable, "-m", "zajit.build", "--only", "no-such-leaf", "--plugins-root", str(ROOT / "tests")], cwd=PKG,
                       capture_output=True, text=True)
    assert r.returncode == 2 and "no leaf matches" in r.stderr


def test_fixture_cases_of_a_leaf():
    from zajit import check
    assert check.cases_of("DDT") == ["DDT_default", "DDT_diffuse_ragged", "DDT_far_extreme", "DDT_near_eco_direct"]
    assert check.cases_of("fx_stft") == ["fx_stft_default"] and check.cases_of("fx_stft4k") == ["fx_stft4k_default"]
    assert check.cases_of("NoSuchLeaf") == []


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", ["DDT", "ERBTilt", "fx_delaytaps"])
def test_correctness_check_rows(leaf):
    """Every (fixture case, kernel) of a leaf within the reference's tolerances; the report carries what the reference's monitor
    shows (max / RMS delta in dBFS, worst variable, worst mem[] cell)."""
    import zabatch
    from zajit import check
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    rows = check.check_leaf(leaf, verbose=False)
    assert {r["path"] for r in rows} == {"fast", "generic"} and len(rows) == 2 * len(check.cases_of(leaf))
    for r in rows:
        assert r["ok"], r
        assert r["max_dbfs"] <= -100.0 and r["vars_worst"] <= 1e-8 and r["mem_high_ok"]
        assert r["mem_worst"] is not None and r["mem_worst"] <= 1e-8          # (these leaves' arenas are small: compared)
    assert any(r["kernel"].startswith("zab_ddt") or r["kernel"].endswith("_tpar") for r in rows if r["path"] == "fast")


@pytest.mark.gpu
def test_correctness_check_fails_loudly(monkeypatch, tmp_path):
    """A fixture that disagrees with the device (here: one recorded output sample moved by 1e-3) makes the check fail."""
    import zabatch
    from zajit import check
    if not zabatch.module_path("fx_dynkat").exists():
        pytest.skip("fx_dynkat not built")
    g = dict(np.load(ROOT / "tests" / "golden" / "fx_dynkat_default.npz", allow_pickle=False))
    g["out"] = g["out"].copy(); g["out"][0, 100] += 1e-3
    np.savez(tmp_path / "fx_dynkat_default.npz", **g)
    monkeypatch.setenv("ZA_GOLDEN_DIR", str(tmp_path))
    rows = check.check_leaf("fx_dynkat", verbose=False)
    assert rows and not any(r["ok"] for r in rows) and all(-61 < r["max_dbfs"] < -59 for r in rows)


def test_hazard_rule_reads_the_process_kernel_and_marks_what_it_rebuilt():
    """zajit/build.py: a module whose generic process kernel sits at the 512-register ceiling with HAZARD_SPILL_BYTES or more of
    spills per lane is compiled again with the GCN pressure trackers (DESIGN.md "Compiler hazards"); the choice is recorded beside
    the module and in its text, so that the next build does not compile twice and a reader can see which build a leaf got."""
    from zajit import build as zb
    lib = zb.LIB
    if not (lib / "libzab_DDT.so").exists():
        pytest.skip("modules not built")
    assert zb.ceiling_spill_bytes(lib / "libzab_DDT.so") < zb.HAZARD_SPILL_BYTES        # (its generic kernel: 512 registers, a few bytes)
    marked = sorted(p.stem for p in lib.glob("*.trackers"))
    for leaf in marked:
        text = (zb.GEN / f"{leaf}_module.hip").read_text()
        assert "-amdgpu-use-amdgpu-trackers=1" in text.rsplit("// leaf build flags:", 1)[-1], leaf
        note = (lib / f"{leaf}.trackers").read_text().split()
        assert len(note) == 2 and int(note[0]) >= zb.HAZARD_SPILL_BYTES, (leaf, note)
    for so in lib.glob("libzab_*.so"):             # nothing the rule would take was left on the default build
        leaf = so.stem[len("libzab_"):]
        if leaf not in marked and leaf not in zb.FAST_KERNELS and (zb.GEN / f"{leaf}_module.hip").exists():
            assert zb.ceiling_spill_bytes(so) < zb.HAZARD_SPILL_BYTES, leaf
