#!/usr/bin/env python3
"""Micro-benchmark: row-panel kernel (sc_proj_ln_proj) against the GEMM /
reduce+LN / GEMM sequence it replaces.  Run under rocprofv3 --kernel-trace to
get per-kernel durations.  Usage (GPU box): python tools/panel_bench.py [rows ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechcatcher_amd.hip_backend import HipBackend

rows = [int(a) for a in sys.argv[1:]] or [16, 160, 1280, 2560, 5120]
be = HipBackend("cuda:0")
ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
be.lib.sc_set_workspace(ws.data_ptr(), ws.numel())
D = 256
for M in rows:
    A = torch.randn(M, D, device="cuda")
    W1, W2 = torch.randn(D, D, device="cuda") / 16, torch.randn(D, D, device="cuda") / 16
    b1, b2 = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    X, XN, Q = torch.zeros(M, D, device="cuda"), torch.zeros(M, D, device="cuda"), torch.zeros(M, D, device="cuda")

    from speechcatcher_amd.weights import pack_lane_weight
    W1p, W2p = pack_lane_weight(W1), pack_lane_weight(W2)

    def panel2():
        be.proj_ln_proj(A, D, W1p, b1, X, D, g, b, None, W2p, b2, Q, M, D)

    def panel1():
        be.proj_ln_proj(A, D, W1p, b1, X, D, g, b, XN, None, None, None, M, D)

    def three():
        be.gemm_ln(A, None, D, W1, b1, X, None, D, M, D, D, g, b, XN, residual=True)
        be.gemm(XN, None, D, W2, b2, Q, None, D, M, D, D)

    def two():
        be.gemm_ln(A, None, D, W1, b1, X, None, D, M, D, D, g, b, XN, residual=True)

    for name, fn in (("panel proj+ln+proj", panel2), ("gemm_ln + gemm", three), ("panel proj+ln", panel1),
                     ("gemm_ln", two)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"M={M:5d} {name:20s} {e0.elapsed_time(e1) * 1e3 / n:8.1f} us/iter (back-to-back launches)")
