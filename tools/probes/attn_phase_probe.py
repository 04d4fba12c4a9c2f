"""Probe (temporary instrumentation build): time the phases of the decoder flash
attention kernel for S active streams (dbg code in bits 8.. of the layer index)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sb, be = bench.build_batch(S, 10, False, 24, "cuda:0")
bench.preload_audio(sb, 24)
bench.run_steps(sb, 20)
torch.cuda.synchronize()
ctrl = np.zeros((S, 8), np.int32)
for s, st in enumerate(sb.st):
    ctrl[s] = [1, st.cur, 0, st.T_ctc, st.L, st.nhyp, 1, 0]
print("L", [st.L for st in sb.st][:8], "T", [st.T_ctc for st in sb.st][:4])
sb.ctrl.copy_(torch.from_numpy(ctrl))
sb.n_rows_step = S * sb.W
torch.cuda.synchronize()


def t(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, fn in (("self", be.dec_self_attn), ("cross", be.dec_cross_attn)):
    out = []
    for dbg in (1, 2, 3, 0):
        out.append(f"dbg{dbg} {t(lambda: fn(sb, 3 | (dbg << 8))):6.2f}")
    print(name, "us/launch (back-to-back):", " | ".join(out), " [1: exit after setup, 2: after row-list build, 3: before merge, 0: full]")
