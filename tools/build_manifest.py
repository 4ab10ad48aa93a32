#!/usr/bin/env python3
"""sha256 of every source file the measured build is made of (the HIP library's sources, the C ABI headers, bench.py and the
Python host layer).  tools/profile_round.sh writes it next to the evidence it collects; tests/test_abi_and_layout.py checks that
the manifest banked under profiles/ matches the tree, i.e. that the committed evidence is of the committed build."""
import hashlib, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
PATTERNS = ("dpdfnet_amd/csrc/*.h", "dpdfnet_amd/csrc/*.hip", "include/*.h", "bench.py", "dpdfnet_amd/*.py")


def manifest() -> dict:
    files = sorted(p for pat in PATTERNS for p in ROOT.glob(pat))
    return {str(p.relative_to(ROOT)): hashlib.sha256(p.read_bytes()).hexdigest() for p in files}


if __name__ == "__main__":
    json.dump({"files": manifest()}, sys.stdout, indent=1)
    print()
