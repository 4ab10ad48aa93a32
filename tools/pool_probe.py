"""Config 5 through the public pool objects with several settings of the pool's knobs (bench.public_streams).
usage: python tools/pool_probe.py [S=64]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for kw in [dict(), dict(spin_s=0.0), dict(spin_s=0.0, regular_window_s=2e-4), dict(spin_s=2e-3), dict(spin_s=1e-3, regular_window_s=5e-3)]:
    for rep in range(2):
        r = bench.public_streams("dpdfnet8_48khz_hr", 48000, S, calls=300, pool_kw=kw)
        print(kw, {k: v for k, v in r.items() if k != "note"}, flush=True)
