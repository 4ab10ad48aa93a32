"""Kernels of the shipped code object against the census of tools/suite_kernel_census.sh: which ones does the GPU suite never launch?
python tools/suite_kernel_census.py [gpurun_out/suite_kernel_census.csv]"""
import csv, shutil, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
census = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "suite_kernel_census.csv"
seen = {r["Name"].replace("void ", ""): int(r["Calls"]) for r in csv.DictReader(open(census))}
with tempfile.TemporaryDirectory() as td:
    lib = Path(td) / "lib.so"; shutil.copy(ROOT / "dpdfnet_amd" / "libdpdfnet_hip.so", lib)
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(lib)], check=True, capture_output=True, cwd=td)
    co = next(Path(td).glob("*gfx950*"))
    sym = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "--wide", str(co)], check=True, capture_output=True, text=True).stdout
names = [l.split()[7] for l in sym.splitlines() if " FUNC " in l and len(l.split()) > 7]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
kernels = sorted({d.replace("void ", "") for d in dem})
dead = [k for k in kernels if k not in seen]
print(f"{len(kernels)} kernels in the code object, {len(kernels) - len(dead)} launched by the suite's own process")
for k in dead: print("  never launched:", k)
