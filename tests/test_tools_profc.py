"""tools/profc (the dynamic per-source-line instruction ledger, DESIGN.md section 6 "Round 6"): the two pieces that run without a GPU --
the IR rewrite of clang's region-counter updates into cn_prof_hit() calls, and the hand-written decoder of the coverage mapping
(this toolchain ships no llvm-cov / llvm-profdata) -- on a toy kernel compiled for gfx950 by the same steps as tools/profc/build.sh."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
LL = "/opt/rocm/lib/llvm/bin/"

TOY = r'''
#include <hip/hip_runtime.h>
template <bool A> __device__ __forceinline__ double f(double x)
{
    if (x > 0.5) return A ? x * 2 : x * 3;
    return -1.0;
}
extern "C" __global__ void k(const double* a, double* o, int n)
{
    int i = threadIdx.x;
    double s = 0;
    for (int j = 0; j < n; ++j) {
        if (a[j] > 0.25) s += f<true>(a[j]) * i;
        else s -= 1.0;
    }
    o[i] = s;
}
'''


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(LL + "lld")), reason="needs the ROCm toolchain")
def test_region_counters_are_rewritten_and_the_coverage_mapping_decodes(tmp_path):
    src = tmp_path / "toy.hip"
    src.write_text(TOY)
    blk = os.path.join(ROOT, "tools", "profc", "profc_block.h")
    ll, llm, obj, out, js = (str(tmp_path / n) for n in ("toy.ll", "toy_m.ll", "toy.o", "toy.out", "toy.json"))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-include", blk, "-fprofile-instr-generate",
                           "-fprofile-update=atomic", "-fcoverage-mapping", "-gline-tables-only", "--cuda-device-only", "-emit-llvm", "-S",
                           "-o", ll, str(src)], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profc", "rewrite_ir.py"), ll, llm], capture_output=True, text=True, check=True)
    # k has 3 counters (body, loop body, then-branch), f<true> has 3: all six updates rewritten, only the helper's own atomics are left
    assert "6 counter updates rewritten" in r.stdout, r.stdout
    text = open(llm).read()
    assert text.count("call void @cn_prof_hit(") == 6 and "atomicrmw add ptr addrspace(1) @__profc_k" not in text
    subprocess.check_call([LL + "clang", "-x", "ir", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O1", "-fPIC", "-c", llm, "-o", obj],
                          stderr=subprocess.DEVNULL)
    subprocess.check_call([LL + "lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", out, obj])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "profc", "covmap.py"), out, js], stdout=subprocess.DEVNULL)
    d = json.load(open(js))
    fn = {f["name"]: f for f in d["functions"]}
    assert {"k", "cn_profc_kernel"} <= set(fn) and any("1fILb1E" in n for n in fn)
    assert d["anchor_before_bytes"] == fn["cn_profc_kernel"]["cnt_off"] * 8
    assert sum(f["ncnt"] for f in d["functions"]) == d["cnts_u64"]
    k = fn["k"]
    assert k["ncnt"] == 3 and len(k["cov"]) == 1
    regs = [r_ for r_ in k["cov"][0]["regions"] if r_["kind"] == "code" and r_["file"] == 0]
    body = regs[0]
    assert body["c"] == ["c", 0] and (body["ls"], body["le"]) == (9, 17)     # the kernel's body: '{' on line 9 to '}' on line 17 of the toy
    loop = [r_ for r_ in regs if r_["c"] == ["c", 1]]
    assert loop and all(body["ls"] <= r_["ls"] <= r_["le"] <= body["le"] for r_ in loop)
    assert any(os.path.basename(p_) == "toy.hip" for p_ in d["filenames"])
