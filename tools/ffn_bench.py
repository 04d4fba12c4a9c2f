#!/usr/bin/env python3
"""Micro-benchmark: fused feed-forward (sc_ffn_ln) against the two GEMMs (+ reduce/LN) it replaces.
Usage (GPU box): python tools/ffn_bench.py [rows ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechcatcher_amd.hip_backend import HipBackend
from speechcatcher_amd.weights import pack_panel_weight, split_panel_weight

rows = [int(a) for a in sys.argv[1:]] or [160, 320, 560, 800, 1280, 2560, 5376]
be = HipBackend("cuda:0")
D, F = 256, 2048
W1, W2 = torch.randn(F, D, device="cuda") / 16, torch.randn(D, F, device="cuda") / 45
b1, b2 = torch.randn(F, device="cuda"), torch.randn(D, device="cuda")
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
W1p, W2p = pack_panel_weight(W1), pack_panel_weight(W2)
W1h, W2h = W1p.half(), W2p.half()
W1s, W2s = split_panel_weight(W1p), split_panel_weight(W2p)
for M in rows:
    XN, X, LN = torch.randn(M, D, device="cuda"), torch.zeros(M, D, device="cuda"), torch.zeros(M, D, device="cuda")
    H = torch.zeros(M, F, device="cuda")

    def fused():
        be.ffn_ln(XN, None, M, D, F, W1p, b1, W2p, b2, X, g, b, LN)

    def two():
        be.gemm(XN, None, D, W1, b1, H, None, F, M, F, D, relu=True)
        be.gemm_ln(H, None, F, W2, b2, X, None, D, M, D, F, g, b, LN, residual=True)

    def fused_h():
        be.ffn_ln_h(XN, None, M, D, F, W1h, b1, W2h, b2, X, g, b, LN)

    def fused_s():
        be.ffn_ln_s(XN, None, M, D, F, W1s, b1, W2s, b2, X, g, b, LN)

    for name, fn in (("fused ffn_ln", fused), ("fused fp16", fused_h), ("fused split16", fused_s), ("gemm + gemm_ln", two)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"M={M:5d} {name:16s} {us:8.1f} us/iter  {4.0 * M * D * F / us / 1e6:7.1f} TFLOP/s")
