"""Randomised soak of the native stream pool (dpdf_streams_submit*): T host threads, each owning a few members, submit random hop counts at
random moments (process / process_many, occasional idle members, members closed and re-opened), for a given time.  Every member's output
must equal an independent single-stream run of the same samples to rounding (the number of streams in a device call selects the kernel
forms: 2e-6 RMS), and no call may hang (watchdog).  usage: python tools/pool_soak.py [seconds=60] [threads=6] [seed=5]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import StreamEnhancer
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kw = dict(model="dpdfnet2", onnx_path="synthetic:20260417")
hop, win = 160, 320
per_thread = 3
pool = StreamEnhancer.pool(T * per_thread, **kw)
stop = time.time() + secs
stats = {"calls": 0, "hops": 0, "reopened": 0}
errors = []
last_progress = [time.time()]
lock = threading.Lock()

def worker(t):
    rng = np.random.default_rng(seed * 1000 + t)
    try:
        members = [pool.enhancer() for _ in range(per_thread)]
        fed = [[] for _ in members]; got = [[] for _ in members]
        def check_and_reopen(i):
            x = np.concatenate(fed[i]) if fed[i] else np.zeros(0, np.float32)
            y = np.concatenate(got[i]) if got[i] else np.zeros(0, np.float32)
            ref_e = StreamEnhancer(**kw)
            ref = ref_e.process(x) if len(x) else np.zeros(0, np.float32)
            if not (y.shape == ref.shape and (len(y) == 0 or float(np.sqrt(np.mean((y - ref) ** 2))) < 2e-6)):
                errors.append(f"thread {t} member {i}: {y.shape} vs {ref.shape}, max diff {float(np.abs(y - ref[:len(y)]).max()) if len(y) and len(ref) >= len(y) else 'n/a'}")
            members[i].close(); members[i] = pool.enhancer(); fed[i] = []; got[i] = []
            with lock: stats["reopened"] += 1
        while time.time() < stop and not errors:
            idx = [i for i in range(per_thread) if rng.random() < 0.8]
            if not idx:
                time.sleep(float(rng.uniform(0, 2e-3))); continue
            if rng.random() < 0.5:      # one process_many over a subset, equal or mixed hop counts
                ks = [int(rng.integers(1, 4)) if rng.random() < 0.3 else 1 for _ in idx]
                chunks = [(0.05 * rng.standard_normal(k * hop + (win - hop if not fed[i] else 0))).astype(np.float32) for i, k in zip(idx, ks)]
                outs = pool.process_many([(members[i], c) for i, c in zip(idx, chunks)])
                for i, c, o in zip(idx, chunks, outs): fed[i].append(c); got[i].append(o)
                n = len(idx)
            else:                       # single-member process() calls
                for i in idx:
                    k = int(rng.integers(1, 3))
                    c = (0.05 * rng.standard_normal(k * hop + (win - hop if not fed[i] else 0))).astype(np.float32)
                    fed[i].append(c); got[i].append(members[i].process(c))
                n = len(idx)
            with lock: stats["calls"] += 1; stats["hops"] += n; last_progress[0] = time.time()
            if rng.random() < 0.02: check_and_reopen(int(rng.integers(0, per_thread)))
            if rng.random() < 0.2: time.sleep(float(rng.uniform(0, 1e-3)))
        for i in range(per_thread): check_and_reopen(i)
        for m_ in members: m_.close()
    except Exception as e:
        errors.append(f"thread {t}: {type(e).__name__}: {e}")

ths = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(T)]
for th in ths: th.start()
hung = False
while any(th.is_alive() for th in ths):
    time.sleep(0.5)
    if time.time() - last_progress[0] > 20.0 and time.time() < stop + 30:
        hung = True; break
    if time.time() > stop + 60: hung = True; break
print({"seconds": secs, "threads": T, **stats, "device_calls": pool.device_calls, "errors": errors[:5], "hung": hung}, flush=True)
os._exit(1 if (errors or hung) else 0)
