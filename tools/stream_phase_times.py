#!/usr/bin/env python3
"""Phase timing of the stream-resident decoder layer kernel (csrc/decoder_stream.hip: dec_layer_stream_kernel) over ALL
workgroups of its last launches - the instrument of tools/layer_phase_times.py (shader-clock stamps of thread 0 of every
workgroup, aligned on the 100 MHz real-time counter) for the round-6 form.
Needs a library built with the stamps:  tools/build_variant.sh phase "-DSC_PHASE_DBG -DSC_PHASE_MIN_GRID=100"
Usage (GPU box): SC_TEST_HOOKS=1 SC_LIB_VARIANT=build_ab/libscasr_phase.so python tools/stream_phase_times.py [streams] [pre-roll chunks] [detail]"""
import ctypes as C
import os
import sys
os.environ["SC_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
PRE = int(sys.argv[2]) if len(sys.argv) > 2 else 36
DETAIL = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = bench.make_weights("cuda:0")
total = PRE + 14
sb = bench.build_native(w, S, 10, False, total)
sb.set_graphs(False)
audio = bench.make_audio(S, total)
bench.roll(sb, audio, PRE)      # (the bench's window: T ~ 600 frames, ~300 tokens at 36)
torch.cuda.synchronize()
f = sb.lib.sc_phase_debug_stream_arm
f.argtypes = [C.c_int]
assert f(-1) == 0
sb.push([(s, audio[s][PRE * bench.CHUNK:(PRE + 1) * bench.CHUNK], False) for s in range(S)])
torch.cuda.synchronize()
names = ["start -> partial sums + x in LDS (+ row list)", "LayerNorm1", "Q|K|V projection (2 passes) + park", "finish + cache append",
         "self walk", "store + merge", "out-proj + residual + LayerNorm2", "q projection + finish", "cross walk", "store + merge",
         "out-proj + residual + LayerNorm3 + stores"]
NWG, RING = 512, 32
buf = (C.c_longlong * (4 * RING * NWG * 16))()
g = sb.lib.sc_phase_debug_stream
g.argtypes = [C.c_void_p]
assert g(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(4, RING, NWG, 16)
nst = len(names) + 1
launches = []
for slot in range(RING):
    t_all = a[0][slot]
    live = (t_all[:, 0] != 0) & (t_all[:, nst - 1] > t_all[:, 0])
    if not live.any():
        continue
    no = np.bincount(t_all[live][:, 12].astype(np.int64)).argmax()
    live &= t_all[:, 12] == no
    launches.append((int(no), t_all[live]))
launches.sort(key=lambda x: x[0])
print(f"dec_layer_stream_kernel, S={S}: {len(launches)} launches in the ring")
for idx, (no, rows) in enumerate(launches):
    t = rows[:, :nst].astype(np.float64)
    rt = rows[:, 14:16].astype(np.float64)
    span_t, span_rt = t[:, -1] - t[:, 0], (rt[:, 1] - rt[:, 0]) / 100.0
    ok = span_rt > 1.0
    tpu = float(np.median(span_t[ok] / span_rt[ok])) if ok.any() else 2400.0
    start = (rt[:, 0] - rt[:, 0].min()) / 100.0
    rel = start[:, None] + (t - t[:, :1]) / tpu
    d = np.diff(t, axis=1) / tpu
    print(f"  launch {no:5d}: grid {int(rows[0, 13]):4d}, {len(rows):3d} workgroups stamped, {tpu:.0f} MHz, span {rel[:, -1].max():6.2f} us, workgroup mean "
          f"{span_t.mean() / tpu:6.2f} max {span_t.max() / tpu:6.2f} | phases (mean): " + " ".join(f"{x:.1f}" for x in d.mean(axis=0)))
    if DETAIL or idx == len(launches) - 1:
        for i, n in enumerate(names):
            print(f"       {n:52s} in phase: mean {d[:, i].mean():6.2f}  max {d[:, i].max():6.2f} us   reached its end at: mean {rel[:, i + 1].mean():6.2f}  "
                  f"first {rel[:, i + 1].min():6.2f}  last {rel[:, i + 1].max():6.2f} us")

# finer stamps inside the merges (kind 1; shader ticks relative to stamp 5 / 9 of the same workgroup are not comparable across
# kinds on the real-time axis, so: differences between consecutive kind-1 stamps, and kind-1 slot 8 / 0 against kind-0 slot 5)
k1 = a[1]
names1 = ["self: partials stored -> behind the barrier + wave sync", "self: fragments requested", "self: merge arithmetic", "self: context stored",
          "cross: (wave sync)", "cross: fragments requested", "cross: merge arithmetic", "cross: context stored"]
for slot in range(RING):
    t0_, t1_ = a[0][slot], k1[slot]
    live = (t1_[:, 0] != 0) & (t1_[:, 3] > t1_[:, 0]) & (t0_[:, 0] != 0)
    if not live.any():
        continue
    r0, r1 = t0_[live].astype(np.float64), t1_[live].astype(np.float64)
    tpu = 2400.0
    print("merge detail (one launch, mean us over workgroups): walk end -> partials stored %.2f | stored -> barrier passed %.2f | fragments requested %.2f | "
          "merge arithmetic %.2f | context stored %.2f | -> behind the workgroup barrier %.2f || cross: walk end -> wave sync %.2f | requested %.2f | arithmetic %.2f | stored %.2f | -> barrier %.2f" % (
              ((r1[:, 8] - r0[:, 5]) / tpu).mean(), ((r1[:, 0] - r1[:, 8]) / tpu).mean(), ((r1[:, 1] - r1[:, 0]) / tpu).mean(),
              ((r1[:, 2] - r1[:, 1]) / tpu).mean(), ((r1[:, 3] - r1[:, 2]) / tpu).mean(), ((r0[:, 6] - r1[:, 3]) / tpu).mean(),
              ((r1[:, 4] - r0[:, 9]) / tpu).mean(), ((r1[:, 5] - r1[:, 4]) / tpu).mean(), ((r1[:, 6] - r1[:, 5]) / tpu).mean(),
              ((r1[:, 7] - r1[:, 6]) / tpu).mean(), ((r0[:, 10] - r1[:, 7]) / tpu).mean()))
    break
