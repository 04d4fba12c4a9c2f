#!/usr/bin/env python3
"""Micro-benchmark: row-tile projections of the encoder layer (sc_rowtile_proj: norm1 + q|k|v Linear, output
Linear + residual + norm2) against the LayerNorm + GEMM launches they replace.
Also: the encoder block attention with the keys of a (block, head) split over 4 waves against one wave per unit.
Usage (GPU box): python tools/rowtile_bench.py [rows ...]"""
import sys
import os
os.environ.setdefault("SC_TEST_HOOKS", "1")   # the library reads its SC_* switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechcatcher_amd.hip_backend import HipBackend
from speechcatcher_amd.weights import pack_panel_weight

rows = [int(a) for a in sys.argv[1:]] or [42, 336, 672, 1344, 2688, 5376, 10752]
be = HipBackend("cuda:0")
D = 256
Wqkv, Wo = torch.randn(3 * D, D, device="cuda") / 16, torch.randn(D, D, device="cuda") / 16
bqkv, bo = torch.randn(3 * D, device="cuda"), torch.randn(D, device="cuda")
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
Wqp, Wop = pack_panel_weight(Wqkv), pack_panel_weight(Wo)


def timeit(fn, n=30):
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
    cases = (
        ("qkv: rowtile (LN folded)", lambda: be.rowtile_proj(X, M, D, Wqp, bqkv, 3 * D, QKV, ln_g=g, ln_b=b), 3),
        ("qkv: layernorm + gemm", lambda: (be.layernorm(X, None, XN, None, M, g, b),
                                           be.gemm(XN, None, D, Wqkv, bqkv, QKV, None, 3 * D, M, 3 * D, D)), 3),
        ("out: rowtile (+res, LN)", lambda: be.rowtile_proj(ATT, M, D, Wop, bo, D, X, R=X, g2=g, b2=b, LN2=XN), 1),
        ("out: gemm + layernorm", lambda: (be.gemm(ATT, None, D, Wo, bo, X, None, D, M, D, D, residual=True),
                                           be.layernorm(X, None, XN, None, M, g, b)), 1),
    )
    for name, fn, nmul in cases:
        us = timeit(fn)
        print(f"M={M:6d} {name:26s} {us:8.1f} us  {2.0 * M * D * D * nmul / us / 1e6:7.1f} TFLOP/s", flush=True)


for nblk in (1, 2, 16, 64, 128, 256):
    R, H = 42, 8
    QKV, ATT = torch.randn(nblk * R, 3 * D, device="cuda"), torch.zeros(nblk * R, D, device="cuda")
    for mode in ("split", "wave"):
        os.environ["SC_ENC_ATTN"] = mode
        us = timeit(lambda: be.enc_attention(QKV, ATT, nblk, R, H, True))
        print(f"enc_attention nblk={nblk:4d} {mode:6s} {us:8.1f} us", flush=True)
