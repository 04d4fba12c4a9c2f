#!/usr/bin/env python3
"""Phase timing of the three small-bucket decoder layer kernels (dec_layer_attn_kernel<self>, <cross>,
ffn_fused_kernel<PRO>): shader real-time-counter stamps (100 MHz) of workgroup (0,0) in their last launch.
Needs a library built with the stamps:  make -C speechcatcher_amd/csrc clean && make -C speechcatcher_amd/csrc EXTRA=-DSC_PHASE_DBG -j8
Usage (GPU box): python tools/layer_phase_times.py [streams]"""
import ctypes as C
import os
import sys
os.environ["SC_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = bench.make_weights("cuda:0")
sb = bench.build_native(w, S, 10, False, 50)
sb.set_graphs(False)
bench.roll(sb, bench.make_audio(S, 50), 36)      # (the bench's window: T ~ 600 frames, ~300 tokens)
torch.cuda.synchronize()
names = {0: ["touch+partials+x", "LayerNorm", "QKV proj MFMA", "split-K reduce+cache append", "attention walk", "merge", "out-proj MFMA", "store"],
         1: ["touch+partials+x", "LayerNorm", "q proj MFMA", "split-K reduce", "attention walk", "merge", "out-proj MFMA", "store"],
         2: ["row ids (round trip)", "residual + head partials -> LDS (round trip)", "barrier", "LayerNorm3", "barrier",
             "W1 fragments still under way", "GEMM1+GEMM2 (cpw chunks)", "store partial"]}
for fn, kinds in (("sc_phase_debug_layer", (0, 1)), ("sc_phase_debug_ffn", (2,))):
    buf = (C.c_longlong * 128)()
    f = getattr(sb.lib, fn)
    f.argtypes = [C.c_void_p]
    assert f(buf) == 0
    for k in kinds:
        t = [buf[k * 32 + i] for i in range(len(names[k]) + 1)]
        d = [(t[i + 1] - t[i]) / 100.0 for i in range(len(names[k]))]
        print(f"kind {k} ({['self', 'cross', 'ffn PRO'][k]}), S={S}: total {sum(d):.2f} us: " + ", ".join(f"{n} {x:.2f}" for n, x in zip(names[k], d)))
