#!/usr/bin/env python3
"""Module-level golden vectors for the Conformer building blocks: the REAL
reference classes (imported read-only) with seeded weights from
speechcatcher_amd.synth.make_conformer_state.  Survey container only."""
import os
import sys
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools" / "ref_shim"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from speechcatcher_amd import synth  # noqa: E402
from speechcatcher.model.attention.multi_head_attention import RelPositionMultiHeadedAttention  # noqa: E402
from speechcatcher.model.layers.convolution import ConvolutionModule  # noqa: E402
from speechcatcher.model.layers.positional_encoding import RelPositionalEncoding  # noqa: E402

out = {}
for name, (C, H, T, B) in {"c64": (64, 4, 50, 2), "c256": (256, 8, 100, 3)}.items():
    conv_sd, att_sd = synth.make_conformer_state(C, H, 31, seed=4321)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, T, C, generator=g)
    cm = ConvolutionModule(C, 31).eval()
    cm.load_state_dict(conv_sd, strict=False)
    att = RelPositionMultiHeadedAttention(H, C).eval()
    att.load_state_dict(att_sd, strict=True)
    rpe = RelPositionalEncoding(C, 0.0).eval()
    with torch.no_grad():
        xs, pe = rpe(x, offset=3)
        out[f"{name}_conv"] = cm(x).numpy()
        out[f"{name}_att"] = att(x, x, x, pe).numpy()
        out[f"{name}_rpe_x"] = xs.numpy()
        out[f"{name}_rpe_pe"] = pe.numpy()
# masked and long-T attention (multi_head_attention.py:366-372): key-padding mask (batch, 1, T), full mask (batch, T, T)
# with one fully masked query row, and T = 300 without a mask
for name, (C, H, T, B) in {"m64": (64, 4, 70, 3), "m256": (256, 8, 150, 2)}.items():
    _, att_sd = synth.make_conformer_state(C, H, 31, seed=4321)
    g = torch.Generator().manual_seed(78)
    x = torch.randn(B, T, C, generator=g)
    att = RelPositionMultiHeadedAttention(H, C).eval()
    att.load_state_dict(att_sd, strict=True)
    rpe = RelPositionalEncoding(C, 0.0).eval()
    lens = [T - 7 * b for b in range(B)]
    kmask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).unsqueeze(1)          # (B, 1, T)
    fmask = torch.tril(torch.ones(T, T, dtype=torch.bool)).unsqueeze(0).repeat(B, 1, 1)    # causal (B, T, T)
    fmask[:, 5, :] = False                                                                  # a fully masked query row
    with torch.no_grad():
        _, pe = rpe(x, offset=0)
        out[f"{name}_att_kmask"] = att(x, x, x, pe, kmask).numpy()
        out[f"{name}_att_fmask"] = att(x, x, x, pe, fmask).numpy()
C, H, T, B = 128, 4, 300, 1
_, att_sd = synth.make_conformer_state(C, H, 31, seed=4321)
x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(79))
att = RelPositionMultiHeadedAttention(H, C).eval()
att.load_state_dict(att_sd, strict=True)
with torch.no_grad():
    _, pe = RelPositionalEncoding(C, 0.0).eval()(x, offset=0)
    out["long300_att"] = att(x, x, x, pe).numpy()
np.savez_compressed(ROOT / "tests" / "golden" / "conformer.npz", **out)
print({k: v.shape for k, v in out.items()})
