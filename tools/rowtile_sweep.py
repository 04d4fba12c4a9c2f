#!/usr/bin/env python3
"""Sweep of the row-tile projection's tile height (16*rtt rows) and column chunks per workgroup (cpw) through
SC_ROWTILE_FORCE, next to the launcher's own choice.  Usage (GPU box): python tools/rowtile_sweep.py [rows ...]"""
import sys
import os
os.environ.setdefault("SC_TEST_HOOKS", "1")   # the library reads its SC_* switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechcatcher_amd.hip_backend import HipBackend
from speechcatcher_amd.weights import pack_panel_weight

rows = [int(a) for a in sys.argv[1:]] or [42, 84, 1344, 2688, 5376]
be = HipBackend("cuda:0")
D = 256
Wqp = pack_panel_weight(torch.randn(3 * D, D, device="cuda") / 16)
Wop = pack_panel_weight(torch.randn(D, D, device="cuda") / 16)
bqkv, bo = torch.randn(3 * D, device="cuda"), torch.randn(D, device="cuda")
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")


def timeit(fn, n=40):
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


for M in rows:
    X, XN, ATT = torch.randn(M, D, device="cuda"), torch.zeros(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    QKV = torch.zeros(M, 3 * D, device="cuda")
    qkv = lambda: be.rowtile_proj(X, M, D, Wqp, bqkv, 3 * D, QKV, ln_g=g, ln_b=b)          # noqa: E731
    out = lambda: be.rowtile_proj(ATT, M, D, Wop, bo, D, XN, R=X, g2=g, b2=b, LN2=XN)      # noqa: E731
    os.environ.pop("SC_ROWTILE_FORCE", None)
    print(f"M={M:5d} launcher's choice: qkv {timeit(qkv):6.1f} us   out {timeit(out):6.1f} us", flush=True)
    for rtt in (1, 2, 3, 4):
        line = f"M={M:5d} rtt={rtt}:"
        for cpw in (1, 2, 3, 6):
            os.environ["SC_ROWTILE_FORCE"] = f"{rtt},{cpw}"
            line += f"  qkv cpw={cpw} {timeit(qkv):6.1f}"
        line += f"   out {timeit(out):6.1f}"
        print(line, flush=True)
