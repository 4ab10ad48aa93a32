"""Static check of the built library (no GPU): no packed-FP32 VALU instruction in any kernel.

DESIGN.md section 6: on gfx950, v_pk_fma_f32 with cross-half operand selection (op_sel) returns a wrong low half in a 16-lane group
when a wave of ANOTHER kernel issues bf16 / f16 MFMAs on the same SIMD at that moment (tools/pk_fma_coissue_probe.hip reproduces it
stand-alone; v_fma_f32 on the same registers at the same time is right).  The engine runs kernels of several streams side by side and
has a bf16-MFMA mode (gru64_limbs), so the whole library is compiled without the packed-FP32 instructions
(-Xclang -target-feature -Xclang -packed-fp32-ops in __graft_entry__.build_hip) -- and this test keeps it that way: it disassembles
the shipped code object and fails on any v_pk_{fma,mul,add}_f32 / v_pk_mov_b32, whoever put it there (compiler flag lost, inline asm)."""
import re
import shutil
import subprocess
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b|\bv_pk_mov_b32\b")


def _disassemble(lib: Path) -> str:
    with tempfile.TemporaryDirectory() as td:
        local = Path(td) / lib.name
        shutil.copy(lib, local)                         # --offloading writes the extracted bundles next to its input
        subprocess.run([str(OBJDUMP), "--offloading", str(local)], check=True, capture_output=True, cwd=td)
        cos = [p for p in Path(td).iterdir() if "amdgcn-amd-amdhsa--gfx950" in p.name]
        assert len(cos) == 1, [p.name for p in Path(td).iterdir()]
        return subprocess.run([str(OBJDUMP), "-d", str(cos[0])], check=True, capture_output=True, text=True).stdout


@pytest.mark.skipif(not OBJDUMP.exists(), reason="llvm-objdump of the ROCm toolchain not present")
def test_shipped_library_has_no_packed_fp32_instructions():
    import __graft_entry__ as ge
    lib = ge.build_hip()
    text = _disassemble(lib)
    kernel, hits = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel = m.group(1)
        elif PACKED.search(line):
            hits.setdefault(kernel, []).append(line.split("//")[0].strip())
    assert sum(1 for l in text.splitlines() if re.match(r"^[0-9a-f]+ <_Z", l)) > 50, "disassembly looks empty"
    assert not hits, "packed-FP32 instructions in: " + "; ".join(f"{k} ({len(v)}x, e.g. {v[0]})" for k, v in list(hits.items())[:8])


@pytest.mark.skipif(not OBJDUMP.exists(), reason="llvm-objdump of the ROCm toolchain not present")
def test_every_kernel_of_the_library_is_launched_by_the_gpu_suite():
    """No dead kernels: profiles/r*_suite_kernel_census.csv is `rocprofv3 --kernel-trace --stats` over `pytest -m gpu` (tools/suite_kernel_census.sh,
    part of the round's collection) reduced to name + calls; every kernel of the shipped code object must be in it.  (Round 6 found five that
    were not -- two 16 kHz instantiations of the two-stage synthesis DFT, a K-split sum, two gemm_rows instantiations behind conditions that
    are never true -- and removed them.)  Skips while the tree runs ahead of the banked evidence, like the manifest test."""
    import csv
    import json
    import sys
    import __graft_entry__ as ge
    banked = sorted((ROOT / "profiles").glob("r*_suite_kernel_census.csv"), key=lambda p: int(re.match(r"r(\d+)_", p.name).group(1)))
    assert banked, "no kernel census banked under profiles/"
    seen = {r["Name"].replace("void ", "") for r in csv.DictReader(open(banked[-1]))}
    lib = ge.build_hip()
    with tempfile.TemporaryDirectory() as td:
        local = Path(td) / lib.name
        shutil.copy(lib, local)
        subprocess.run([str(OBJDUMP), "--offloading", str(local)], check=True, capture_output=True, cwd=td)
        co = next(p for p in Path(td).iterdir() if "amdgcn-amd-amdhsa--gfx950" in p.name)
        sym = subprocess.run([str(OBJDUMP.with_name("llvm-readelf")), "-s", "--wide", str(co)], check=True, capture_output=True, text=True).stdout
    mangled = [l.split()[7] for l in sym.splitlines() if " FUNC " in l and len(l.split()) > 7]
    names = subprocess.run(["c++filt"], input="\n".join(mangled), check=True, capture_output=True, text=True).stdout.splitlines()
    kernels = sorted({n.replace("void ", "") for n in names})
    assert len(kernels) > 50, kernels
    dead = [k for k in kernels if k not in seen]
    if dead:
        sys.path.insert(0, str(ROOT / "tools"))
        import build_manifest
        stale = json.loads((ROOT / "profiles" / "build_manifest.json").read_text())["files"] != build_manifest.manifest()
        if stale:
            pytest.skip(f"profiles/ is of an earlier build; kernels not in its census: {dead}")
    assert not dead, f"kernels the GPU suite never launches (remove them, or add the test that does, then re-run tools/collect_round.sh): {dead}"
