"""The header-only C++ mirror (include/sw_host.hpp) compiles against the C ABI and, on a GPU box, returns the
same values as the oracle for KAT-1 and a MapRef call."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cstdio>
#include "sw_host.hpp"
int main() {
    try {
        sw::Context ctx(0);
        auto r = sw::SmithWaterman::OptAlignments(ctx).call({"ACGT", "CG"});
        std::printf("%d %zu %d %s %s\n", r.first, r.second.size(), r.second[0].first, r.second[0].second[0].c_str(), r.second[0].second[1].c_str());
        std::vector<std::string> reads{"AACA"};
        auto t = sw::Distribution::CombineReadsToRef().call({{">gi|x", "AAAA"}}, reads, {{1, -1, -1}, sw::ALIGN_TYPES});
        auto m = sw::Distribution::MapRef(ctx).call(t[0]);
        std::printf("%d %zu", m.first, m.second.second.size());
        for (auto &s : m.second.second) std::printf(" %d:%s/%s", s.first, s.second[0].c_str(), s.second[1].c_str());
        std::printf("\n");
    } catch (const sw::Error &e) { std::printf("ERROR %s\n", e.what()); return 3; }
    return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "mirror.cpp"
    src.write_text(SRC)
    exe = tmp_path / "mirror"
    lib = os.path.join(ROOT, "sparksmithwaterman_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", lib, "-lswmi",
                           "-Wl,-rpath," + lib, "-o", str(exe)])
    return exe


def test_cpp_mirror_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stdout


@pytest.mark.gpu
def test_cpp_mirror_values(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([str(exe)], capture_output=True, text=True, check=True)
    lines = p.stdout.strip().splitlines()
    assert lines[0] == "10 1 2 CG CG"                                                     # KAT-1
    assert lines[1] == "2 5 1:AA/AA 1:AA_A/AACA 1:AAAA/AACA 2:AA/AA 3:AA/AA"              # KAT-3 after MapRef's stable sort
