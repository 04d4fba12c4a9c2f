#!/usr/bin/env python3
"""Phase timing of the persistent stream-cluster decoder kernel (csrc/decoder_cluster.hip): shader-clock stamps of
workgroup (first stream, head 0) at the phase boundaries of the last launch.  Usage (GPU box): python tools/cluster_phase_times.py [streams]"""
import ctypes as C
import os
import sys
os.environ["SC_TEST_HOOKS"] = "1"
os.environ["SC_CLUSTER_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = bench.make_weights("cuda:0")
sb = bench.build_native(w, S, 10, False, 30)
sb.set_graphs(False)
bench.preload_audio(sb, 30)
bench.run_steps(sb, 12)
buf = (C.c_longlong * 256)()
n = sb.lib.sc_dec_cluster_debug(buf, 256)
names = ["reduce+LN1", "qkv proj", "qkv reduce", "self attn", "outproj1", "barrier A", "reduce+LN2", "q proj+reduce", "cross attn",
         "outproj2+barrier B", "reduce+LN3", "FFN gemm1", "FFN gemm2", "barrier C"]
t = [buf[i] for i in range(n)]
per = 15
print(f"{n} stamps; shader clock ticks -> us at 100 MHz? (s_memtime counts at a constant 100 MHz on gfx9)")
for li in (0, 1, 7, 13):
    seg = t[li * per:(li + 1) * per]
    if len(seg) < per:
        break
    d = [(seg[i + 1] - seg[i]) for i in range(per - 1)]
    print(f"layer {li}: total {sum(d)} ticks: " + ", ".join(f"{nm} {x}" for nm, x in zip(names, d)))
print("whole kernel (14 layers):", t[14 * per - 1] - t[0] if n >= 14 * per else "n/a", "ticks")
