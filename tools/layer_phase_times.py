#!/usr/bin/env python3
"""Phase timing of the three decoder layer kernels (dec_layer_attn_kernel<self>, <cross>, ffn_fused_kernel<PRO>) over ALL
workgroups of their last 32 launches: shader-clock stamps of thread 0 of every workgroup (common.h: SC_STAMP), aligned on
the 100 MHz real-time counter (the shader clocks of the eight XCDs are not synchronised).
Per launch: span (first start -> last end), mean / max workgroup time, and per phase the mean / max time a workgroup spends in it.
Needs a library built with the stamps:  tools/build_variant.sh phase "-DSC_PHASE_DBG -DSC_PHASE_MIN_GRID=200"
Usage (GPU box): python tools/layer_phase_times.py [streams] [pre-roll chunks] [mode] [detail]
  mode: "chunk" (default) = after the lock-step pre-roll ONE more chunk is pushed with all launches stamped: the ring keeps the
        last 32 launches of each kind, i.e. the layers of the chunk's last decode steps at full grids (SC_PHASE_MIN_GRID);
        "first" = only the 28 launches of its first two decode steps;  "serve" = a continuous-batching window (sc_submit / sc_poll)
  detail: 1 = the phase table of every launch, 0 = of the last one only"""
import ctypes as C
import os
import sys
os.environ["SC_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PRE = int(sys.argv[2]) if len(sys.argv) > 2 else 36
MODE = sys.argv[3] if len(sys.argv) > 3 else "chunk"
DETAIL = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w = bench.make_weights("cuda:0")
total = PRE + 14
sb = bench.build_native(w, S, 10, False, total)
sb.set_graphs(False)
audio = bench.make_audio(S, total)
bench.roll(sb, audio, PRE)      # (the bench's window: T ~ 600 frames, ~300 tokens at 36)
torch.cuda.synchronize()
for fn in ("sc_phase_debug_layer_arm", "sc_phase_debug_ffn_arm"):
    f = getattr(sb.lib, fn)
    f.argtypes = [C.c_int]
    assert f(28 if MODE == "first" else -1) == 0
if MODE == "serve":
    a3 = audio.reshape(S, -1, bench.CHUNK)
    nxt = np.full(S, PRE, np.int64)
    r = bench.serve(sb, a3, nxt, 4, max(1, S // 8))
    print(f"continuous window: {r['elapsed'] / 4 * 1e3:.2f} ms per step, {r['iterations_per_step']:.1f} iterations per step")
else:
    sb.push([(s, audio[s][PRE * bench.CHUNK:(PRE + 1) * bench.CHUNK], False) for s in range(S)])
torch.cuda.synchronize()
names = {0: ["start -> partial sums + x in LDS (+ row list)", "LayerNorm", "QKV proj MFMA (+ first K|V batch requested)", "split-K reduce + cache append", "attention walk", "merge", "out-proj MFMA", "store"],
         1: ["start -> partial sums + x in LDS", "LayerNorm", "q proj MFMA", "split-K reduce", "attention walk", "merge", "out-proj MFMA", "store"],
         2: ["row ids (round trip)", "residual + head partials -> LDS (round trip)", "barrier", "LayerNorm3", "barrier",
             "W1 fragments still under way", "GEMM1+GEMM2 (cpw chunks)", "store partial"]}
NWG, RING = 512, 32
for fn, kinds in (("sc_phase_debug_layer", (0, 1)), ("sc_phase_debug_ffn", (2,))):
    buf = (C.c_longlong * (4 * RING * NWG * 16))()
    f = getattr(sb.lib, fn)
    f.argtypes = [C.c_void_p]
    assert f(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(4, RING, NWG, 16)
    for k in kinds:
        nst = len(names[k]) + 1
        launches = []
        for slot in range(RING):
            t_all = a[k][slot]
            live = (t_all[:, 0] != 0) & (t_all[:, nst - 1] > t_all[:, 0])
            if not live.any():
                continue
            no = np.bincount(t_all[live][:, 12].astype(np.int64)).argmax()      # (a slot holds ONE launch: its number)
            live &= t_all[:, 12] == no
            launches.append((int(no), t_all[live]))
        launches.sort(key=lambda x: x[0])
        print(f"kind {k} ({['self', 'cross', 'ffn PRO'][k]}), S={S}: {len(launches)} launches in the ring")
        for idx, (no, rows) in enumerate(launches):
            t = rows[:, :nst].astype(np.float64)
            rt = rows[:, 14:16].astype(np.float64)
            span_t, span_rt = t[:, -1] - t[:, 0], (rt[:, 1] - rt[:, 0]) / 100.0
            ok = span_rt > 1.0
            tpu = float(np.median(span_t[ok] / span_rt[ok])) if ok.any() else 2400.0     # shader ticks per microsecond
            start = (rt[:, 0] - rt[:, 0].min()) / 100.0
            rel = start[:, None] + (t - t[:, :1]) / tpu
            d = np.diff(t, axis=1) / tpu
            print(f"  launch {no:5d}: grid {int(rows[0, 13]):4d}, {len(rows):3d} workgroups stamped, {tpu:.0f} MHz, span {rel[:, -1].max():6.2f} us, workgroup mean "
                  f"{span_t.mean() / tpu:6.2f} max {span_t.max() / tpu:6.2f}, starts within {start.max():.2f} us | phases (mean): " +
                  " ".join(f"{x:.1f}" for x in d.mean(axis=0)))
            extra = rows[:, 9:12].astype(np.float64)
            if os.environ.get("SC_PROJ_STAMPS") and (extra > 0).all():
                print("       projection: wave 0 done | youngest wave done | behind the barrier, after the LayerNorm's barrier: " +
                      " ".join(f"{x:.2f}" for x in ((extra - t[:, 2:3]) / tpu).mean(axis=0)) + " us")
            elif k == 0 and (extra > 0).all():   # SC_SELF_ROLES: when the waves of roles 1..3 (head group 0) finished their projection
                print("       projection by role, done after the LayerNorm's barrier (role 1 = q, K quarters 2-3 | 2 = k | 3 = v): " +
                      " ".join(f"{x:.2f}" for x in ((extra - t[:, 2:3]) / tpu).mean(axis=0)) + " us")
            if DETAIL or idx == len(launches) - 1:
                for i, n in enumerate(names[k]):
                    print(f"       {n:52s} in phase: mean {d[:, i].mean():6.2f}  max {d[:, i].max():6.2f} us   reached its end at: mean {rel[:, i + 1].mean():6.2f}  "
                          f"first {rel[:, i + 1].min():6.2f}  last {rel[:, i + 1].max():6.2f} us")
