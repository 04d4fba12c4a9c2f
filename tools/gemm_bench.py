#!/usr/bin/env python3
"""Micro-benchmark of sc_gemm on the GEMM shapes of the hot path (fp32 MFMA).
Usage (GPU box): python tools/gemm_bench.py [streams]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechcatcher_amd.hip_backend import HipBackend

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
be = HipBackend("cuda:0")
shapes = [
    ("enc qkv", S * 42, 768, 256, {}), ("enc out+res", S * 42, 256, 256, {"residual": True}),
    ("enc ffn1", S * 42, 2048, 256, {"relu": True}), ("enc ffn2+res", S * 42, 256, 2048, {"residual": True}),
    ("conv2 (dense A)", S * 285, 256, 2304, {"relu": True}), ("sub out", S * 15, 256, 4864, {}),
    ("dec qkv", S * 10, 768, 256, {}), ("dec proj", S * 10, 256, 256, {"residual": True}),
    ("dec ffn1", S * 10, 2048, 256, {"relu": True}), ("dec ffn2", S * 10, 256, 2048, {"residual": True}),
    ("dec out", S * 10, 1024, 256, {}), ("ctc/kv rows", S * 16, 512, 256, {}),
]
print(f"streams={S}")
for name, M, N, K, kw in shapes:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        be.gemm(A, None, K, W, b, Cm, None, N, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        be.gemm(A, None, K, W, b, Cm, None, N, M, N, K, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:5d}  {us:9.1f} us  {tf:7.2f} TFLOP/s")
