"""DESIGN.md section 6 hazard: what did df_apply CONSUME when the output was wrong?

Probe build only (build_ab/lib_probe.so = the library with -DDPDF_HAZARD_PROBE; DPDFNET_HIP_LIB points at it).  df_apply_probe_kernel<TAPS>
reads the ten deep-filter taps of a (clip, frame, bin) in one of six ways and STORES the values it consumed (+ XCC id + s_memtime) to a
side buffer behind everything else it does.  Reference = the same limb kernels, serial schedule (overlap 0): bit-identical when clean.
Every record that differs from the reference is classified dword by dword:
  stale1/2/3  = the value the SAME buffer slot held one/two/three chunks earlier (w.coefs is reused per chunk: a line that was never
                invalidated / re-fetched), prev_t/next_t = the neighbouring frame's record, prev_f/next_f = the neighbouring bin's record
                (an address or split problem), zero, poison (0xffffffff: never written) or other.

usage: python tools/hazard_probe.py <runs> <taps modes, e.g. 1,4,5,2,3,0> [uncached=0|1] [limbs=3] [dump=1]
"""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dpdfnet_amd import backend as be
from dpdfnet_amd.weights import synth_blob

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
modes = [int(v.replace("a", "10").replace("b", "11")) for v in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
uncached = int(sys.argv[3]) if len(sys.argv) > 3 else 0
limbs = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dump_on = int(sys.argv[5]) if len(sys.argv) > 5 else 1
CH = int(os.environ.get("PROBE_CHUNK", "64"))

sr, nb = 16000, 4
m = be.HipModel(sr, nb, synth_blob(be.manifest(sr, nb), 20260417), 0)
L = m._L
L.dpdf_probe_dump_alloc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
L.dpdf_probe_dump_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rng = np.random.default_rng(3)
B, NCH = 256, 8
n = 160 * CH * NCH
T = 1 + (n + 320) // 160
D = 96
wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
m.set_chunk_frames(CH)
m.set_option("gru64_limbs", limbs)
if uncached:
    m.set_option("probe_coefs_uncached", 1)


def run(taps, overlap, dump):
    m.set_overlap(overlap)
    # "12" = taps 2 with wait form 1 (one full wait behind the history loads), "22": + idle cycles; "1002": taps 2 + the output recomputed at the very
    # end from the same registers; wait forms 10, 11 are written "a2", "b2" on the command line
    m.set_option("probe_taps", taps % 10)
    m.set_option("probe_wait", (taps // 10) % 100)
    m.set_option("probe_late", taps // 1000)
    m.set_option("probe_dump", 1 if dump else 0)
    if dump:
        assert L.dpdf_probe_dump_alloc(m._h, B, T) == 0
    y = m.enhance_batch(wav, None)
    d = None
    if dump:
        d = np.empty((B, T, 161, 36), np.uint32)
        assert L.dpdf_probe_dump_fetch(m._h, d.ctypes.data, d.size) == 0
    return y, d


F = 161
y_ref, d_ref = run(int(os.environ.get("PROBE_REF_TAPS", "1000")), 0, True)            # serial schedule: the reference
R = d_ref[..., :22].copy(); R_late = d_ref[..., 24:26].copy(); del d_ref
print(f"reference (limbs={limbs}, serial): T={T} frames, dump poison rows: {(R[..., 20:22] == 0xffffffff).all(axis=-1).sum()} of {B * T * F}", flush=True)
GROUPS = {"taps": slice(0, 10), "hist": slice(10, 20), "out": slice(20, 22)}

snap_tot = collections.Counter()
for taps in modes:
    grp_hist = collections.Counter(); kinds = collections.Counter(); pos_in_chunk = collections.Counter(); xcc_bad = collections.Counter()
    fhist = collections.Counter(); chunk_hist = collections.Counter(); nhist = collections.Counter(); wave_only = 0
    bad_runs = bad_clips_tot = bad_rec_tot = 0
    explain = collections.Counter(); examples = []
    for r in range(runs):
        y, d = run(taps, 27, dump_on)
        e = np.abs(y - y_ref).reshape(B, -1, 160).max(axis=2)
        bad_clips = np.nonzero(e.max(axis=1) > 1e-5)[0]
        bad_clips_tot += len(bad_clips); bad_runs += len(bad_clips) > 0
        if not dump_on:
            continue
        C = d[..., :22]
        neq = C != R
        late_ok = (d[..., 24:26] == R_late).all(axis=-1)
        if (taps // 10) % 100 in (3, 4, 5, 8):          # history copies taken right behind the (staged) waits vs the same registers at the end of the kernel
            sn = d[:, :, :D, 26:36] != d[:, :, :D, 10:20]
            nsn = int(sn.any(axis=-1).sum())
            snap_tot[taps] += nsn
            if nsn:
                w = np.argwhere(sn)
                which = collections.Counter(int(j) // 2 for _, _, _, j in w[:5000])
                kinds_s = collections.Counter()
                for b, t, f, j in w[:3000]:
                    v = d[b, t, f, 26 + j]
                    k = "other"
                    if v == 0: k = "zero"
                    else:
                        for n2 in range(5):        # the value another of the five loads delivered (same lane): registers shared with an address / another destination
                            if v == d[b, t, f, 10 + 2 * n2 + (j & 1)] and n2 != j // 2: k = f"value of load {n2}"; break
                        else:
                            if t >= CH and v == R[b, t - CH, f, 10 + j]: k = "same slot one chunk earlier"
                    kinds_s[k] += 1
                print(f"    run {r}: COPY BEHIND THE WAIT != REGISTER AT THE END in {nsn} records (f < D); load index {dict(which)}; what the copy held: {dict(kinds_s)}; "
                      f"first {w[:3].tolist()}", flush=True)
        if (taps // 10) % 100 == 11:           # scalar twins beside the packed products: dumped (x 1, not x 1/wnorm) in the first copy slot
            tw = d[:, :, :D, 26:28].view(np.float32) * np.float32(320.0)
            pk = d[:, :, :D, 20:22].view(np.float32); ref_o = R[:, :, :D, 20:22].view(np.float32)
            pk_bad = (pk != ref_o).any(axis=-1); tw_bad = (np.abs(tw - ref_o) > 1e-5 * np.abs(ref_o).max()).any(axis=-1)
            print(f"    run {r}: packed sums wrong in {int(pk_bad.sum())} records, scalar twins wrong in {int(tw_bad.sum())}, both in {int((pk_bad & tw_bad).sum())}", flush=True)
        rec = np.argwhere(neq.any(axis=-1))          # [k][b, t, f]
        bad_rec_tot += len(rec)
        rec_clips = set(rec[:, 0].tolist())
        wave_only += sum(1 for b in bad_clips if b not in rec_clips)      # waveform bad, everything df_apply consumed AND produced clean
        for b, t, f in rec[:20000]:
            tc = t % CH
            pos_in_chunk[int(tc)] += 1; fhist[int(f) // 8 * 8] += 1; chunk_hist[int(t // CH)] += 1
            xcc_bad[int(d[b, t, f, 22]) & 0xf] += 1
            grp_hist["+".join(k for k, sl in GROUPS.items() if neq[b, t, f, sl].any()) + ("|late recompute right" if late_ok[b, t, f] else "|late recompute wrong too")] += 1
            for j in np.nonzero(neq[b, t, f, 10:20])[0]:
                n = j // 2; nhist[int(n)] += 1
                v = C[b, t, f, 10 + j]
                kind = "other"
                for k in (1, 2, 3):
                    if t - k * CH >= 0 and v == R[b, t - k * CH, f, 10 + j]: kind = f"stale{k}(same slot, {k} chunks ago)"; break
                else:
                    if v == 0: kind = "zero"
                    elif v == 0xffffffff: kind = "poison"
                    else:
                        for dt in (-4, -3, -2, -1, 1, 2, 3, 4):
                            if 0 <= t + dt < T and v == R[b, t + dt, f, 10 + j]: kind = f"frame{dt:+d}"; break
                kinds[kind] += 1
        if len(rec) and os.environ.get("PROBE_EXPLAIN", "1") != "0":
            # what would give the wrong output from the RIGHT inputs?  re = sum_n (s_n.x c_n.r - s_n.y c_n.i), im = sum_n (s_n.x c_n.i + s_n.y c_n.r), x 1/wnorm
            inv = 320.0
            for b, t, f in rec[:3000]:
                tp = d[b, t, f, 0:10].view(np.float32).astype(np.float64); hs = d[b, t, f, 10:20].view(np.float32).astype(np.float64)
                got = d[b, t, f, 20:22].view(np.float32).astype(np.float64) / inv; want = R[b, t, f, 20:22].view(np.float32).astype(np.float64) / inv
                rr = hs[0::2] * tp[0::2]; ii = hs[1::2] * tp[1::2]; ri = hs[0::2] * tp[1::2]; ir = hs[1::2] * tp[0::2]
                tol = 1e-5 * max(1e-6, np.abs(np.concatenate([rr, ii, ri, ir])).max())
                wrong = ("re" if abs(got[0] - want[0]) > tol else "") + ("im" if abs(got[1] - want[1]) > tol else "")
                expl = "unexplained"
                # candidates: one product term missing / doubled, a prefix of the accumulation lost (sum starts at term k)
                def close(x, y): return abs(x - y) <= tol
                for k in range(1, 5):
                    if close(got[0], (rr[k:].sum() - ii[k:].sum())) and close(got[1], (ri[k:].sum() + ir[k:].sum())): expl = f"both sums lost terms 0..{k - 1}"
                    elif close(got[0], rr[k:].sum() - ii.sum()) : expl = f"rr chain lost terms 0..{k - 1}"
                    elif close(got[0], rr.sum() - ii[k:].sum()) : expl = f"ii chain lost terms 0..{k - 1}"
                    elif close(got[1], ri[k:].sum() + ir.sum()) : expl = f"ri chain lost terms 0..{k - 1}"
                    elif close(got[1], ri.sum() + ir[k:].sum()) : expl = f"ir chain lost terms 0..{k - 1}"
                for n2 in range(5):
                    if close(got[0], want[0] - rr[n2]) or close(got[0], want[0] + ii[n2]) or close(got[1], want[1] - ri[n2]) or close(got[1], want[1] - ir[n2]): expl = f"one product of term {n2} missing"
                if close(got[0], 0) and close(got[1], 0): expl = "zero"
                explain[wrong + ": " + expl] += 1
                if len(examples) < 6: examples.append((int(b), int(t), int(f), wrong, expl, got.tolist(), want.tolist(), rr.tolist(), ii.tolist(), ri.tolist(), ir.tolist()))
        if len(bad_clips) or len(rec):
            frames = sorted({(int(b), int(t)) for b, t, _ in rec[:2000]})
            bclip_frames = [(int(b), sorted(set((np.nonzero(e[b] > 1e-5)[0]).tolist()))[:4]) for b in bad_clips[:4]]
            print(f"  taps={taps} run {r}: {len(bad_clips)} bad clips {bclip_frames}, {len(rec)} bad records in {len({(b, t) for b, t, _ in rec})} frames; e.g. {frames[:6]}", flush=True)
    print(f"taps={taps} uncached={uncached} limbs={limbs} dump={dump_on}: {runs} runs, {bad_runs} with bad clips ({bad_clips_tot} clips), "
          f"{bad_rec_tot} bad dump records; bad clips whose df_apply inputs AND outputs are all clean: {wave_only}", flush=True)
    if (taps // 10) % 100 in (3, 4, 5, 8):
        print(f"   copies behind the waits that differ from the final registers: {snap_tot[taps]} records over {runs} runs")
    if bad_rec_tot:
        print("   what the wrong output equals:", dict(explain))
        for ex in examples: print("     e.g. (b, t, f) =", ex[:3], ex[3], "|", ex[4], "| got", ex[5], "want", ex[6], "\n        rr", ex[7], "\n        ii", ex[8], "\n        ri", ex[9], "\n        ir", ex[10])
        print("   groups differing   :", dict(grp_hist))
        print("   history word kinds :", dict(kinds))
        print("   history tap n      :", dict(sorted(nhist.items())))
        print("   frame pos in chunk :", dict(sorted(pos_in_chunk.items())))
        print("   chunk index        :", dict(sorted(chunk_hist.items())))
        print("   bins f (by 8)      :", dict(sorted(fhist.items())))
        print("   consumer XCC       :", dict(sorted(xcc_bad.items())), flush=True)
