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
