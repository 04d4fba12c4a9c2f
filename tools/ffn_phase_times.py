#!/usr/bin/env python3
"""Phase timing of the prologue-less fused feed-forward (ffn_fused_kernel<.., PRO = false>: the encoder layers' feed-forward and,
round 6, the stream-resident decoder form's) over all workgroups of its last launches - stamps of thread 0 of every workgroup
(kind 3 of csrc/gemm.hip: start | row tile in LDS | both GEMMs of all chunks done | partial sums stored).
Needs a library built with the stamps: tools/build_variant.sh phase "-DSC_PHASE_DBG -DSC_PHASE_MIN_GRID=100"
Usage (GPU box): SC_TEST_HOOKS=1 SC_LIB_VARIANT=build_ab/libscasr_phase.so python tools/ffn_phase_times.py [streams=48] [pre-roll=30]"""
import ctypes as C
import os
import sys
os.environ["SC_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
PRE = int(sys.argv[2]) if len(sys.argv) > 2 else 30
DETAIL = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = bench.make_weights("cuda:0")
total = PRE + 14
sb = bench.build_native(w, S, 10, False, total)
sb.set_graphs(False)
audio = bench.make_audio(S, total)
bench.roll(sb, audio, PRE)
torch.cuda.synchronize()
f = sb.lib.sc_phase_debug_ffn_arm
f.argtypes = [C.c_int]
assert f(-1) == 0
sb.push([(s, audio[s][PRE * bench.CHUNK:(PRE + 1) * bench.CHUNK], False) for s in range(S)])
torch.cuda.synchronize()
names = ["row tile -> LDS", "GEMM 1 + GEMM 2 of all chunks", "partial sums -> LDS -> memory"]
NWG, RING = 512, 32
buf = (C.c_longlong * (4 * RING * NWG * 16))()
g = sb.lib.sc_phase_debug_ffn
g.argtypes = [C.c_void_p]
assert g(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(4, RING, NWG, 16)
nst = len(names) + 1
for slot in range(RING):
    t_all = a[3][slot]
    live = (t_all[:, 0] != 0) & (t_all[:, nst - 1] > t_all[:, 0])
    if not live.any():
        continue
    no = np.bincount(t_all[live][:, 12].astype(np.int64)).argmax()
    rows = t_all[live & (t_all[:, 12] == no)]
    t = rows[:, :nst].astype(np.float64)
    rt = rows[:, 14:16].astype(np.float64)
    span_t, span_rt = t[:, -1] - t[:, 0], (rt[:, 1] - rt[:, 0]) / 100.0
    ok = span_rt > 1.0
    tpu = float(np.median(span_t[ok] / span_rt[ok])) if ok.any() else 2400.0
    start = (rt[:, 0] - rt[:, 0].min()) / 100.0
    rel = start[:, None] + (t - t[:, :1]) / tpu
    d = np.diff(t, axis=1) / tpu
    print(f"launch {no:4d}: grid {int(rows[0, 13]):4d}, {len(rows):3d} workgroups, span {rel[:, -1].max():6.2f} us, workgroup mean {span_t.mean() / tpu:6.2f} max {span_t.max() / tpu:6.2f}, "
          f"starts within {start.max():.2f} us | phases (mean / max): " + " | ".join(f"{n}: {d[:, i].mean():.2f} / {d[:, i].max():.2f}" for i, n in enumerate(names)))
    if slot == 0 or DETAIL:
        raw = rows[:, :12].astype(np.float64)
        if (raw[:, 10] > 0).all():
            seq = [0, 1, 4, 5, 6, 7, 8, 9, 10, 2, 3]
            dd = np.diff(raw[:, seq], axis=1) / tpu
            lab = ["tile->LDS", "GEMM1 c0", "relu+LDS+barrier", "GEMM2 c0", "GEMM1 c1", "relu+barriers", "GEMM2 c1", "(to stamp 10)", "stage", "store"]
            print("   inner (mean us): " + " | ".join(f"{l} {dd[:, i].mean():.2f}" for i, l in enumerate(lab)))
        wg = np.nonzero(live & (t_all[:, 12] == no))[0]
        order = np.argsort(start)
        print("   start time (us) by workgroup id, every 16th:", " ".join(f"{w}:{start[i]:.1f}" for i, w in enumerate(wg) if w % 16 == 0))
        print("   start time by XCD (id % 8): " + " ".join(f"x{x}: {start[wg % 8 == x].min():.1f}-{start[wg % 8 == x].max():.1f}" for x in range(8)))
