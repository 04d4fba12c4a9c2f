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
