#!/usr/bin/env python3
"""Sweep the (tile height, chunks per workgroup) choices of the fused FFN for the
row counts the compaction buckets produce; prints time per call for each forced
configuration (env SC_FFN_FORCE="rtt,cpw") next to the cost model's choice.
Usage (GPU box): python tools/ffn_sweep.py [f32|fp16|split16]"""
import os
os.environ.setdefault("SC_TEST_HOOKS", "1")   # the library reads its SC_* switches only with this set
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORM = next((a for a in sys.argv[1:] if a in ("f32", "fp16", "split16")), "f32")
ROWS = (160, 320, 640, 800, 960, 1040, 1120, 1200, 1280, 2688, 2940, 5376)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from speechcatcher_amd.hip_backend import HipBackend
    from speechcatcher_amd.weights import pack_panel_weight, split_panel_weight
    be = HipBackend("cuda:0")
    D, F = 256, 2048
    W1, W2 = torch.randn(F, D, device="cuda") / 16, torch.randn(D, F, device="cuda") / 45
    b1, b2 = torch.randn(F, device="cuda"), torch.randn(D, device="cuda")
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    W1p, W2p = pack_panel_weight(W1), pack_panel_weight(W2)
    fn = be.ffn_ln
    if FORM == "fp16":
        W1p, W2p, fn = W1p.half(), W2p.half(), be.ffn_ln_h
    elif FORM == "split16":
        W1p, W2p, fn = split_panel_weight(W1p), split_panel_weight(W2p), be.ffn_ln_s
    out = []
    for M in ROWS:
        XN, X, LN = torch.randn(M, D, device="cuda"), torch.zeros(M, D, device="cuda"), torch.zeros(M, D, device="cuda")
        for _ in range(5):
            fn(XN, None, M, D, F, W1p, b1, W2p, b2, X, g, b, LN)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn(XN, None, M, D, F, W1p, b1, W2p, b2, X, g, b, LN)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{e0.elapsed_time(e1) * 1e3 / 40:7.1f}")
    print(" ".join(out))
    sys.exit(0)

print(f"weight form {FORM}")
print("rows:            " + " ".join(f"{m:7d}" for m in ROWS))
for force in ["model"] + [f"{r},{c}" for r in ((1, 2, 3) if FORM == "split16" else (1, 2, 3, 4, 5)) for c in (1, 2, 4, 8)]:
    env = dict(os.environ)
    if force != "model":
        env["SC_FFN_FORCE"] = force
    r = subprocess.run([sys.executable, __file__, "child", FORM], env=env, capture_output=True, text=True)
    print(f"rtt,cpw={force:6s}   " + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-200:]))
