"""CPU-side checks of the drop-in boundary: libswmi.so loads and exports exactly what include/swmi.h declares.
No compute call is made here (there is no GPU in the build container and no CPU fallback in the library)."""
import os
import re

import pytest

import sparksmithwaterman_amd as sw
from sparksmithwaterman_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "swmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swmi_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_bound_and_exported():
    names = _declared()
    assert len(names) >= 18
    bound = {n for n, _, _ in _capi.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)
    lib = _capi.load()
    for n in names:
        assert getattr(lib, n) is not None


def test_abi_version_and_default_params():
    lib = _capi.load()
    assert lib.swmi_abi_version() == 3
    p = _capi.Params()
    lib.swmi_default_params(p)
    assert (p.match, p.mismatch, p.gap, p.tie_mode, p.types) == (5, -3, -4, 0, b"aid-")


def test_make_params_mirrors_java_arrays():
    p = sw.make_params([1, -1, -2], ["x", "y", "z", "."], sw.TIE_STRICT)
    assert (p.match, p.mismatch, p.gap, p.tie_mode, p.types) == (1, -1, -2, 1, b"xyz.")
    with pytest.raises(ValueError):
        sw.make_params([1, 2])


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sw.SwmiError) as e:
        sw.Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sparksmithwaterman_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "sw_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_generated_instruction_streams_are_what_the_generators_emit():
    """swmi_step_gen.inc / swmi_cells_gen.inc are committed: a change to tools/gen_step.py or tools/gen_cells.py (or a hand edit
    of the .inc) that is not regenerated would ship a kernel the scripts' hazard checks never saw."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("SWMI_GEN_DOT8C", None)
    for script in ("gen_step.py", "gen_cells.py"):
        p = subprocess.run([sys.executable, os.path.join(root, "tools", script), "--check"], capture_output=True, text=True, env=env)
        assert p.returncode == 0, (script, p.stdout[-300:], p.stderr[-300:])
