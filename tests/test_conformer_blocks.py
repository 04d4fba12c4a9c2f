"""Conformer building blocks: module-level parity against
golden vectors produced by the REAL reference classes (tools/gen_golden_conformer.py)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from speechcatcher_amd import conformer, synth
from speechcatcher_amd.mel import positional_encoding_table

CASES = {"c64": (64, 4, 50, 2), "c256": (256, 8, 100, 3)}


def _inputs(name):
    C, H, T, B = CASES[name]
    conv_sd, att_sd = synth.make_conformer_state(C, H, 31, seed=4321)
    g = torch.Generator().manual_seed(77)
    return C, H, T, B, conv_sd, att_sd, torch.randn(B, T, C, generator=g)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_and_spec_match_reference_modules(name):
    from oracle.kernel_spec import SpecBackend
    from oracle.ref_port import conformer_conv_module, conformer_relpos_mha
    gold = np.load(GOLDEN / "conformer.npz")
    C, H, T, B, conv_sd, att_sd, x = _inputs(name)
    pe_tab = positional_encoding_table(5000, C)
    xs, pe = conformer.rel_positional_encoding(pe_tab, x, offset=3)
    np.testing.assert_array_equal(xs.numpy(), gold[f"{name}_rpe_x"])
    np.testing.assert_array_equal(pe.unsqueeze(0).numpy(), gold[f"{name}_rpe_pe"])
    np.testing.assert_allclose(conformer_conv_module(conv_sd, x).numpy(), gold[f"{name}_conv"], atol=1e-5)
    np.testing.assert_allclose(conformer_relpos_mha(att_sd, H, x, pe.unsqueeze(0)).numpy(), gold[f"{name}_att"], atol=1e-5)
    be = SpecBackend()
    out = conformer.conv_module(be, conformer.pack_conv_module(conv_sd, "cpu"), x)
    np.testing.assert_allclose(out.numpy(), gold[f"{name}_conv"], atol=2e-5)
    out = conformer.relpos_mha(be, conformer.pack_relpos_mha(att_sd, "cpu"), x, pe)
    np.testing.assert_allclose(out.numpy(), gold[f"{name}_att"], atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_conformer_blocks_match_reference_modules(name):
    from speechcatcher_amd.hip_backend import HipBackend
    be = HipBackend("cuda:0")
    gold = np.load(GOLDEN / "conformer.npz")
    C, H, T, B, conv_sd, att_sd, x = _inputs(name)
    pe = positional_encoding_table(5000, C)[3:3 + T]
    out = conformer.conv_module(be, conformer.pack_conv_module(conv_sd, "cuda:0"), x.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), gold[f"{name}_conv"], atol=1e-4, rtol=1e-4)
    out = conformer.relpos_mha(be, conformer.pack_relpos_mha(att_sd, "cuda:0"), x.cuda(), pe.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), gold[f"{name}_att"], atol=1e-4, rtol=1e-4)


# masked and long-T attention (multi_head_attention.py:366-372) - inputs as tools/gen_golden_conformer.py builds them
MASKED = {"m64": (64, 4, 70, 3), "m256": (256, 8, 150, 2)}


def _masked_inputs(name):
    C, H, T, B = MASKED[name]
    _, att_sd = synth.make_conformer_state(C, H, 31, seed=4321)
    x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(78))
    lens = [T - 7 * b for b in range(B)]
    kmask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).unsqueeze(1)
    fmask = torch.tril(torch.ones(T, T, dtype=torch.bool)).unsqueeze(0).repeat(B, 1, 1)
    fmask[:, 5, :] = False
    return C, H, T, B, att_sd, x, kmask, fmask


def _long_inputs():
    C, H, T = 128, 4, 300
    _, att_sd = synth.make_conformer_state(C, H, 31, seed=4321)
    return C, H, T, att_sd, torch.randn(1, T, C, generator=torch.Generator().manual_seed(79))


def run_masked_attention(be, device, atol, rtol=0.0):
    gold = np.load(GOLDEN / "conformer.npz")
    for name in MASKED:
        C, H, T, B, att_sd, x, kmask, fmask = _masked_inputs(name)
        pe = positional_encoding_table(5000, C)[:T].to(device)
        w = conformer.pack_relpos_mha(att_sd, device)
        for key, mask in (("kmask", kmask), ("fmask", fmask)):
            out = conformer.relpos_mha(be, w, x.to(device), pe, mask=mask)
            np.testing.assert_allclose(out.cpu().numpy(), gold[f"{name}_att_{key}"], atol=atol, rtol=rtol, err_msg=f"{name} {key}")
    C, H, T, att_sd, x = _long_inputs()
    out = conformer.relpos_mha(be, conformer.pack_relpos_mha(att_sd, device), x.to(device),
                               positional_encoding_table(5000, C)[:T].to(device))
    np.testing.assert_allclose(out.cpu().numpy(), gold["long300_att"], atol=atol, rtol=rtol)


def test_oracle_and_spec_masked_and_long_attention_match_reference_module():
    from oracle.kernel_spec import SpecBackend
    from oracle.ref_port import conformer_relpos_mha
    gold = np.load(GOLDEN / "conformer.npz")
    for name in MASKED:
        C, H, T, B, att_sd, x, kmask, fmask = _masked_inputs(name)
        pe = positional_encoding_table(5000, C)[:T].unsqueeze(0)
        np.testing.assert_allclose(conformer_relpos_mha(att_sd, H, x, pe, kmask).numpy(), gold[f"{name}_att_kmask"], atol=1e-5)
        np.testing.assert_allclose(conformer_relpos_mha(att_sd, H, x, pe, fmask).numpy(), gold[f"{name}_att_fmask"], atol=1e-5)
    run_masked_attention(SpecBackend(), "cpu", atol=2e-5)


@pytest.mark.gpu
def test_hip_masked_and_long_attention_match_reference_module():
    from speechcatcher_amd.hip_backend import HipBackend
    run_masked_attention(HipBackend("cuda:0"), "cuda:0", atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
def test_hip_tiled_attention_equals_single_tile_kernel_and_handles_ragged_masks():
    """the tiled kernel with an all-ones mask == the LDS-resident kernel (T <= 128); random (B, T, T) masks with
    empty rows against the spec; T not a multiple of the key tile or the query tile"""
    from oracle.kernel_spec import SpecBackend
    from speechcatcher_amd.hip_backend import HipBackend
    be, sp = HipBackend("cuda:0"), SpecBackend()
    g = torch.Generator().manual_seed(5)
    for (B, T, H, dk) in ((2, 100, 4, 32), (1, 129, 2, 16), (2, 257, 4, 64), (1, 1, 4, 32), (3, 64, 8, 32)):
        d = H * dk
        qkv = torch.randn(B * T, 3 * d, generator=g)
        p = torch.randn(T, d, generator=g)
        bu, bv = torch.randn(H, dk, generator=g) * 0.3, torch.randn(H, dk, generator=g) * 0.3
        masks = [None, torch.ones(B, T, dtype=torch.uint8), (torch.rand(B, T, generator=g) > 0.3).to(torch.uint8),
                 (torch.rand(B, T, T, generator=g) > 0.5).to(torch.uint8)]
        masks[3][:, T // 2, :] = 0
        outs = []
        for mask in masks:
            want = torch.zeros(B * T, d)
            sp.relpos_attention(qkv, p, bu, bv, want, B, T, H, mask=mask)
            got = torch.full((B * T, d), float("nan"), device="cuda:0")
            be.relpos_attention(qkv.cuda(), p.cuda(), bu.cuda(), bv.cuda(), got, B, T, H,
                                mask=None if mask is None else mask.cuda())
            torch.cuda.synchronize()
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-5, rtol=1e-4, err_msg=f"{(B, T, H, dk)}")
            outs.append(got.cpu())
        np.testing.assert_allclose(outs[1].numpy(), outs[0].numpy(), atol=2e-6)
        assert torch.all(outs[3].view(B, T, d)[:, T // 2] == 0)
