"""GPU parity tests proper: every HIP kernel against its torch spec, through
the C ABI (ctypes).  Run with `pytest -m gpu` on an MI355X."""
import json

import numpy as np
import pytest
import torch

from conftest import load_case
from speechcatcher_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from speechcatcher_amd.hip_backend import HipBackend
    return HipBackend("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


GEMM_SHAPES = [
    # M, N, K  (decoder rows, encoder rows, tiny model, ragged edges)
    (10, 256, 256), (42, 768, 256), (42, 2048, 256), (42, 256, 2048), (1280, 1024, 256),
    (5376, 2048, 256), (5376, 256, 2048), (285, 256, 2304), (16, 256, 4864), (7, 192, 64),
    (100, 64, 576), (33, 1024, 64), (129, 130, 96), (1, 64, 32),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("mode", ["plain", "relu", "residual"])
def test_gemm_matches_torch(hip, M, N, K, mode):
    from oracle.kernel_spec import SpecBackend
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    C0 = _rand(M, N, seed=4)
    ref = C0.clone()
    SpecBackend().gemm(A, None, K, W, b, ref, None, N, M, N, K, relu=(mode == "relu"), residual=(mode == "residual"))
    for naive in (False, True):
        Cg = C0.cuda()
        hip.gemm(A.cuda(), None, K, W.cuda(), b.cuda(), Cg, None, N, M, N, K, relu=(mode == "relu"),
                 residual=(mode == "residual"), naive=naive)
        torch.cuda.synchronize()
        np.testing.assert_allclose(Cg.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=2e-4, err_msg=f"naive={naive}")


def test_gemm_mfma_is_k_ordered_fma_chain(hip):
    """The f32 MFMA is an exact fp32 fma chain: the tiled kernel must agree
    bit-for-bit with the scalar fmaf kernel (same k order)."""
    M, N, K = 300, 320, 256   # K < 512: this shape never takes the split-K path
    A, W, b = _rand(M, K, seed=5).cuda(), _rand(N, K, seed=6).cuda(), _rand(N, seed=7).cuda()
    C1, C2 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    hip.gemm(A, None, K, W, b, C1, None, N, M, N, K)
    hip.gemm(A, None, K, W, b, C2, None, N, M, N, K, naive=True)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)


def test_gemm_row_tables_and_implicit_conv(hip):
    from oracle.kernel_spec import SpecBackend
    d, F1, F2, T1 = 64, 39, 19, 9
    T2 = (T1 - 3) // 2 + 1
    c1 = _rand(T1 * F1 + 5, d, seed=8)
    tt, ff = np.meshgrid(np.arange(T2), np.arange(F2), indexing="ij")
    a_rows = torch.from_numpy(((2 * tt) * F1 + 2 * ff).reshape(-1).astype(np.int32))
    M = a_rows.numel()
    W, b = _rand(d, 9 * d, seed=9, scale=0.05), _rand(d, seed=10)
    c_rows = torch.from_numpy(np.random.RandomState(0).permutation(M + 7)[:M].astype(np.int32))
    ref = torch.zeros(M + 7, d)
    SpecBackend().gemm(c1, a_rows, d, W, b, ref, c_rows, d, M, d, 9 * d, relu=True, conv_f1=F1)
    out = torch.zeros(M + 7, d, device="cuda")
    hip.gemm(c1.cuda(), a_rows.cuda(), d, W.cuda(), b.cuda(), out, c_rows.cuda(), d, M, d, 9 * d, relu=True, conv_f1=F1)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=2e-4)
    # and against torch's own conv2d on the (C, T, F) view
    x = c1[:T1 * F1].view(T1, F1, d).permute(2, 0, 1).unsqueeze(0)
    wconv = W.view(d, 3, 3, d).permute(0, 3, 1, 2).contiguous()
    y = torch.relu(torch.nn.functional.conv2d(x, wconv, b, stride=2))[0].permute(1, 2, 0).reshape(M, d)
    np.testing.assert_allclose(out.cpu()[c_rows.long()].numpy(), y.numpy(), atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("M,N,K,mode", [(300, 320, 256, "plain"), (2100, 256, 4864, "plain"), (1100, 1024, 256, "relu"),
                                        (4000, 512, 256, "residual"), (129, 130, 96, "plain"), (700, 64, 2304, "relu")])
def test_gemm_split16(hip, M, N, K, mode):
    """SC_GEMM_SPLIT16: the tiled GEMM with both operands split into fp16 hi + lo / 2^11 when the tiles are staged
    (three v_mfma_f32_32x32x16_f16 per 16 k values, fp32 accumulation): against float64 as close as the f32-MFMA kernel
    (a small multiple of its error), rows of very different magnitude, with a gather / scatter row table, split-K shapes
    included (K = 4864: the subsampling Linear)."""
    A, W, b = _rand(M + 9, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    A *= torch.logspace(-3, 2, M + 9)[:, None]
    C0 = _rand(M + 9, N, seed=4)
    a_rows = torch.randperm(M + 9, generator=torch.Generator().manual_seed(3))[:M].to(torch.int32)
    c_rows = torch.randperm(M + 9, generator=torch.Generator().manual_seed(4))[:M].to(torch.int32)
    y = A[a_rows.long()].double() @ W.double().t() + b.double()
    scale = A[a_rows.long()].double().abs() @ W.double().abs().t() + b.double().abs()
    if mode == "relu":
        y = torch.relu(y)
    ref = C0.double().clone()
    ref[c_rows.long()] = (C0[c_rows.long()].double() if mode == "residual" else 0.0) + y
    if mode == "residual":
        scale = scale + C0[c_rows.long()].double().abs()
    err = {}
    for form in ("f32", "split"):
        Cg = C0.cuda()
        hip.gemm(A.cuda(), a_rows.cuda(), K, W.cuda(), b.cuda(), Cg, c_rows.cuda(), N, M, N, K, relu=(mode == "relu"),
                 residual=(mode == "residual"), split16=(form == "split"))
        torch.cuda.synchronize()
        out = Cg.cpu()
        untouched = torch.ones(M + 9, dtype=torch.bool)
        untouched[c_rows.long()] = False
        assert torch.equal(out[untouched], C0[untouched])
        err[form] = float(((out[c_rows.long()].double() - ref[c_rows.long()]).abs() / scale).max())
    assert err["f32"] < 2e-6 and err["split"] < 2e-6, err
    assert err["split"] < 4.0 * err["f32"] + 2e-7, err


def test_gemm_split16_implicit_conv(hip):
    """... and the second Conv2d of the subsampling as an implicit GEMM (row gather + tap offsets) in the split form,
    against torch's conv2d."""
    d, F1, F2, T1 = 256, 39, 19, 33
    T2 = (T1 - 3) // 2 + 1
    c1 = torch.relu(_rand(T1 * F1 + 5, d, seed=8))
    tt, ff = np.meshgrid(np.arange(T2), np.arange(F2), indexing="ij")
    a_rows = torch.from_numpy(((2 * tt) * F1 + 2 * ff).reshape(-1).astype(np.int32))
    M = a_rows.numel()
    W, b = _rand(d, 9 * d, seed=9, scale=0.02), _rand(d, seed=10)
    out = torch.zeros(M, d, device="cuda")
    hip.gemm(c1.cuda(), a_rows.cuda(), d, W.cuda(), b.cuda(), out, None, d, M, d, 9 * d, relu=True, conv_f1=F1, split16=True)
    torch.cuda.synchronize()
    x = c1[:T1 * F1].view(T1, F1, d).permute(2, 0, 1).unsqueeze(0).double()
    wconv = W.view(d, 3, 3, d).permute(0, 3, 1, 2).contiguous().double()
    y = torch.relu(torch.nn.functional.conv2d(x, wconv, b.double(), stride=2))[0].permute(1, 2, 0).reshape(M, d)
    np.testing.assert_allclose(out.cpu().numpy(), y.float().numpy(), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("M,N,K", [(10, 256, 256), (10, 256, 2048), (42, 64, 64), (1280, 256, 256), (1280, 256, 2048), (300, 256, 512)])
def test_gemm_ln_fused(hip, M, N, K):
    """GEMM + residual + LayerNorm fused into the split-K reduce epilogue."""
    from oracle.kernel_spec import SpecBackend
    A, W, b = _rand(M, K, seed=21), _rand(N, K, seed=22, scale=K ** -0.5), _rand(N, seed=23)
    g, be_ = 1 + 0.1 * _rand(N, seed=24), _rand(N, seed=25)
    C0 = _rand(M, N, seed=26)
    refC, refL = C0.clone(), torch.zeros(M, N)
    SpecBackend().gemm_ln(A, None, K, W, b, refC, None, N, M, N, K, g, be_, refL, residual=True)
    Cg, Lg = C0.cuda(), torch.zeros(M, N, device="cuda")
    hip.gemm_ln(A.cuda(), None, K, W.cuda(), b.cuda(), Cg, None, N, M, N, K, g.cuda(), be_.cuda(), Lg, residual=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Cg.cpu().numpy(), refC.numpy(), atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(Lg.cpu().numpy(), refL.numpy(), atol=3e-4, rtol=3e-4)


@pytest.mark.parametrize("M,D", [(10, 256), (16, 256), (1280, 256), (1275, 256), (1030, 256), (2101, 256), (50, 128), (23, 64), (160, 64)])
@pytest.mark.parametrize("second", [True, False])
def test_proj_ln_proj_row_panel(hip, M, D, second):
    """Row-panel kernel (out-projection + residual + LayerNorm + next projection,
    v_mfma_f32_4x4x1_16B_f32, 4/8/16-row panels) against gemm_ln + gemm of the spec; ragged last panel."""
    from oracle.kernel_spec import SpecBackend
    A, W1, b1 = _rand(M, D, seed=31), _rand(D, D, seed=32, scale=D ** -0.5), _rand(D, seed=33)
    W2, b2 = _rand(D, D, seed=34, scale=D ** -0.5), _rand(D, seed=35)
    g, be_ = 1 + 0.1 * _rand(D, seed=36), _rand(D, seed=37)
    X0 = _rand(M, D, seed=38)
    refX, refN, refQ = X0.clone(), torch.zeros(M, D), torch.zeros(M, D)
    from speechcatcher_amd.weights import pack_lane_weight, unpack_lane_weight
    W1p, W2p = pack_lane_weight(W1), pack_lane_weight(W2)
    assert torch.equal(unpack_lane_weight(W1p), W1)
    # the device-side packer of the C ABI produces the same permutation
    dev_p = torch.empty(D, D, device="cuda")
    assert hip.lib.sc_pack_lane_weight(W1.cuda().data_ptr(), D, D, dev_p.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dev_p.cpu(), W1p)
    SpecBackend().proj_ln_proj(A, D, W1p, b1, refX, D, g, be_, refN, W2p if second else None, b2 if second else None,
                               refQ if second else None, M, D)
    Xg = X0.cuda()
    Ng = torch.zeros(M + 3, D, device="cuda")   # rows past M must stay untouched
    Qg = torch.zeros(M + 3, D, device="cuda")
    hip.proj_ln_proj(A.cuda(), D, W1p.cuda(), b1.cuda(), Xg, D, g.cuda(), be_.cuda(), Ng if not second else None,
                     W2p.cuda() if second else None, b2.cuda() if second else None, Qg if second else None, M, D)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Xg.cpu().numpy(), refX.numpy(), atol=2e-4, rtol=2e-4)
    if second:
        np.testing.assert_allclose(Qg[:M].cpu().numpy(), refQ.numpy(), atol=3e-4, rtol=3e-4)
        assert float(Qg[M:].abs().max()) == 0.0
    else:
        np.testing.assert_allclose(Ng[:M].cpu().numpy(), refN.numpy(), atol=3e-4, rtol=3e-4)
        assert float(Ng[M:].abs().max()) == 0.0


@pytest.mark.parametrize("M,D,F", [(10, 256, 2048), (1280, 256, 2048), (533, 256, 2048), (5376, 256, 2048),
                                   (77, 128, 256), (200, 256, 128)])
@pytest.mark.parametrize("with_ln", [True, False])
def test_ffn_fused(hip, M, D, F, with_ln):
    """Fused feed-forward (hidden activations in LDS, 16x16x4 f32 MFMA, partials reduced
    in fixed order) against gemm + gemm_ln of the spec; with and without a row table."""
    from oracle.kernel_spec import SpecBackend
    from speechcatcher_amd.weights import pack_panel_weight
    XN, X0, L0 = _rand(M + 5, D, seed=61), _rand(M + 5, D, seed=62), _rand(M + 5, D, seed=63)
    W1, b1 = _rand(F, D, seed=64, scale=D ** -0.5), _rand(F, seed=65)
    W2, b2 = _rand(D, F, seed=66, scale=F ** -0.5), _rand(D, seed=67)
    g, be_ = 1 + 0.1 * _rand(D, seed=68), _rand(D, seed=69)
    W1p, W2p = pack_panel_weight(W1), pack_panel_weight(W2)
    dev_p = torch.empty(F, D, device="cuda")   # device-side packer, rectangular
    assert hip.lib.sc_pack_panel_weight(W1.cuda().data_ptr(), F, D, dev_p.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dev_p.cpu(), W1p)
    spec = SpecBackend()
    for rows in (None, torch.randperm(M + 5, generator=torch.Generator().manual_seed(5))[:M].to(torch.int32)):
        refX, refL = X0.clone(), L0.clone()
        spec.ffn_ln(XN, rows, M, D, F, W1p, b1, W2p, b2, refX, g, be_, refL if with_ln else None)
        Xg, Lg = X0.cuda(), L0.cuda()
        hip.ffn_ln(XN.cuda(), None if rows is None else rows.cuda(), M, D, F, W1p.cuda(), b1.cuda(), W2p.cuda(),
                   b2.cuda(), Xg, g.cuda(), be_.cuda(), Lg if with_ln else None)
        torch.cuda.synchronize()
        np.testing.assert_allclose(Xg.cpu().numpy(), refX.numpy(), atol=3e-4, rtol=3e-4)
        np.testing.assert_allclose(Lg.cpu().numpy(), refL.numpy(), atol=5e-4, rtol=5e-4)


@pytest.mark.parametrize("M,D,F", [(80, 256, 2048), (1280, 256, 2048), (5376, 256, 2048), (37, 128, 512), (640, 256, 2048)])
def test_ffn_fused_fp16_weights(hip, M, D, F):
    """The fused feed-forward with fp16 weights (sc_ffn_ln_h: v_mfma_f32_16x16x16_f16, activations rounded to fp16
    when staged, fp32 accumulation / bias / residual / LayerNorm) against the same arithmetic in torch: operands
    rounded to fp16, products and sums in fp32."""
    from speechcatcher_amd.weights import pack_panel_weight
    XN, X0, L0 = _rand(M + 5, D, seed=61), _rand(M + 5, D, seed=62), _rand(M + 5, D, seed=63)
    W1, b1 = _rand(F, D, seed=64, scale=D ** -0.5), _rand(F, seed=65)
    W2, b2 = _rand(D, F, seed=66, scale=F ** -0.5), _rand(D, seed=67)
    g, be_ = 1 + 0.1 * _rand(D, seed=68), _rand(D, seed=69)
    W1h, W2h = pack_panel_weight(W1).half(), pack_panel_weight(W2).half()
    r16 = lambda t: t.half().float()   # noqa: E731
    for rows in (None, torch.randperm(M + 5, generator=torch.Generator().manual_seed(5))[:M].to(torch.int32)):
        idx = torch.arange(M) if rows is None else rows.long()
        h = torch.relu(r16(XN[idx]).double() @ r16(W1).double().t() + b1.double()).float()
        y = (r16(h).double() @ r16(W2).double().t()).float() + b2
        refX = X0.clone()
        refX[idx] = X0[idx] + y
        refL = L0.clone()
        refL[idx] = torch.nn.functional.layer_norm(refX[idx], (D,), g, be_, 1e-12)
        Xg, Lg = X0.cuda(), L0.cuda()
        hip.ffn_ln_h(XN.cuda(), None if rows is None else rows.cuda(), M, D, F, W1h.cuda(), b1.cuda(), W2h.cuda(),
                     b2.cuda(), Xg, g.cuda(), be_.cuda(), Lg)
        torch.cuda.synchronize()
        # the hidden activations are rounded to fp16 after an fp32 sum whose order differs from torch's: a value on a
        # rounding boundary may land on the neighbouring fp16 (2^-11 relative of ONE of 2048 terms)
        np.testing.assert_allclose(Xg.cpu().numpy(), refX.numpy(), atol=2e-3, rtol=1e-3)
        np.testing.assert_allclose(Lg.cpu().numpy(), refL.numpy(), atol=2e-3, rtol=1e-3)
        # ... and against the fp32 kernel: the fp16 rounding of the operands shows (not the same numbers), bounded
        Xf = X0.cuda()
        hip.ffn_ln(XN.cuda(), None if rows is None else rows.cuda(), M, D, F, pack_panel_weight(W1).cuda(), b1.cuda(),
                   pack_panel_weight(W2).cuda(), b2.cuda(), Xf, g.cuda(), be_.cuda(), None)
        torch.cuda.synchronize()
        diff = float((Xf - Xg).abs().max())
        assert 0.0 < diff < 3e-2, diff


@pytest.mark.parametrize("M,D,F", [(80, 256, 2048), (1280, 256, 2048), (5376, 256, 2048), (37, 128, 512), (640, 256, 2048),
                                   (10, 256, 2048)])
def test_ffn_fused_split_weights(hip, M, D, F):
    """The fused feed-forward on the fp16 matrix pipe with fp32-grade results (sc_ffn_ln_s: every operand split into
    fp16 hi + lo / 2^11, three v_mfma_f32_16x16x32_f16 per product sum, fp32 accumulation): against float64 it must
    be as close as the fp32 kernel is (a small multiple of its error, nowhere near fp16's 2^-11), over operand
    magnitudes from 1e-3 to 1e2 - and the offline split of the weights must reproduce them to 2^-21 (+ 2e-11)."""
    from speechcatcher_amd.weights import pack_panel_weight, split_panel_weight
    XN, X0, L0 = _rand(M + 5, D, seed=61), _rand(M + 5, D, seed=62), _rand(M + 5, D, seed=63)
    XN *= torch.logspace(-3, 2, M + 5)[:, None]            # rows of very different magnitude
    W1, b1 = _rand(F, D, seed=64, scale=D ** -0.5), _rand(F, seed=65)
    W2, b2 = _rand(D, F, seed=66, scale=F ** -0.5), _rand(D, seed=67)
    W1[:, ::7] *= 1e-3                                     # columns of small weights (fp16-subnormal low parts)
    g, be_ = 1 + 0.1 * _rand(D, seed=68), _rand(D, seed=69)
    W1p, W2p = pack_panel_weight(W1), pack_panel_weight(W2)
    W1s, W2s = split_panel_weight(W1p), split_panel_weight(W2p)
    assert W1s.dtype == torch.float16 and W1s.shape == (F, 2 * D)
    v = W1s.reshape(-1, 2, 64, 8).float()
    back = v[:, 0] + v[:, 1] / 2048.0                      # [n][lane][8] = the two fp32 slabs side by side
    orig = torch.cat([W1p.reshape(-1, 2, 64, 4)[:, 0], W1p.reshape(-1, 2, 64, 4)[:, 1]], dim=-1)
    # 22 significant bits, with an absolute floor where the low part is an fp16 subnormal (|w| < ~6e-5: 2^-25 / 2^11)
    assert bool(((back - orig).abs() <= 2.0 ** -21 * orig.abs() + 2e-11).all())
    for rows in (None, torch.randperm(M + 5, generator=torch.Generator().manual_seed(5))[:M].to(torch.int32)):
        idx = torch.arange(M) if rows is None else rows.long()
        h = torch.relu(XN[idx].double() @ W1.double().t() + b1.double())
        y = h @ W2.double().t() + b2.double()
        ref = X0.double().clone()
        ref[idx] = X0[idx].double() + y
        scale = (h.abs() @ W2.double().abs().t() + X0[idx].double().abs()).clamp_min(1e-3)   # size of the summed terms
        out = {}
        for form in ("f32", "split"):
            Xg, Lg = X0.cuda(), L0.cuda()
            if form == "f32":
                hip.ffn_ln(XN.cuda(), None if rows is None else rows.cuda(), M, D, F, W1p.cuda(), b1.cuda(), W2p.cuda(),
                           b2.cuda(), Xg, g.cuda(), be_.cuda(), Lg)
            else:
                hip.ffn_ln_s(XN.cuda(), None if rows is None else rows.cuda(), M, D, F, W1s.cuda(), b1.cuda(), W2s.cuda(),
                             b2.cuda(), Xg, g.cuda(), be_.cuda(), Lg)
            torch.cuda.synchronize()
            out[form] = (Xg.cpu(), Lg.cpu())
        e32 = float(((out["f32"][0][idx].double() - ref[idx]).abs() / scale).max())
        esp = float(((out["split"][0][idx].double() - ref[idx]).abs() / scale).max())
        assert e32 < 2e-6 and esp < 2e-6, (e32, esp)       # relative to the terms of the sums: fp32 class
        assert esp < 4.0 * e32 + 2e-7, (e32, esp)
        untouched = torch.ones(M + 5, dtype=torch.bool)
        untouched[idx] = False
        assert torch.equal(out["split"][0][untouched], X0[untouched])
        np.testing.assert_allclose(out["split"][1].numpy(), out["f32"][1].numpy(), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("H,dk,R,nblk,masked", [(8, 32, 42, 5, True), (8, 32, 42, 130, True), (8, 32, 7, 3, False),
                                                (8, 32, 1, 2, False), (8, 32, 2, 2, True), (8, 32, 64, 2, False),
                                                (4, 16, 42, 3, True), (4, 64, 42, 3, True)])
@pytest.mark.parametrize("kernel", ["default", "split", "wave"])
def test_enc_attention(hip, monkeypatch, H, dk, R, nblk, masked, kernel):
    """Encoder block attention (multi_head_attention.py:92-133 with the mask of
    contextual_block_transformer_encoder.py:524-528): the default form (d_k = 32, R <= 48: on the matrix cores, round 5;
    otherwise one of the next two), the 4-waves-per-(block, head) kernel with the keys split over the waves, and the
    one-wave-per-unit kernel it replaced (still used for d_k = 64)."""
    from oracle.kernel_spec import SpecBackend
    if kernel == "default":
        monkeypatch.delenv("SC_ENC_ATTN", raising=False)
    else:
        monkeypatch.setenv("SC_ENC_ATTN", kernel)
    d = H * dk
    qkv = _rand(nblk * R, 3 * d, seed=201)
    qkv[:, :d] *= 3.0   # peaked softmax rows as well
    ref = torch.zeros(nblk * R, d)
    SpecBackend().enc_attention(qkv, ref, nblk, R, H, masked)
    out = torch.full((nblk * R + 2, d), 3.0, device="cuda")
    hip.enc_attention(qkv.cuda(), out, nblk, R, H, masked)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[:nblk * R].cpu().numpy(), ref.numpy(), atol=5e-5, rtol=5e-5)
    assert float(out[nblk * R:].min()) == 3.0 and float(out[nblk * R:].max()) == 3.0


@pytest.mark.parametrize("M,D", [(42, 256), (5376, 256), (1000, 256), (333, 128)])
def test_rowtile_proj(hip, M, D):
    """Encoder attention projections with the LayerNorms folded in (row tiles in LDS, 16x16x4 f32 MFMA,
    fragment-packed weights): norm1 + q|k|v Linear, and output Linear + residual (in place) + norm2."""
    from oracle.kernel_spec import SpecBackend
    from speechcatcher_amd.weights import pack_panel_weight
    N = 3 * D
    X, ATT = _rand(M, D, seed=91), _rand(M, D, seed=92)
    Wqkv, bqkv = _rand(N, D, seed=93, scale=D ** -0.5), _rand(N, seed=94)
    Wo, bo = _rand(D, D, seed=95, scale=D ** -0.5), _rand(D, seed=96)
    g1, b1, g2, b2 = 1 + 0.1 * _rand(D, seed=97), _rand(D, seed=98), 1 + 0.1 * _rand(D, seed=99), _rand(D, seed=100)
    Wqp, Wop = pack_panel_weight(Wqkv), pack_panel_weight(Wo)
    spec = SpecBackend()
    refQ = torch.full((M + 3, N), 5.0)
    spec.rowtile_proj(X, M, D, Wqp, bqkv, N, refQ, ln_g=g1, ln_b=b1)
    Qg = torch.full((M + 3, N), 5.0, device="cuda")
    hip.rowtile_proj(X.cuda(), M, D, Wqp.cuda(), bqkv.cuda(), N, Qg, ln_g=g1.cuda(), ln_b=b1.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(Qg.cpu().numpy(), refQ.numpy(), atol=3e-4, rtol=3e-4)   # rows >= M untouched (5.0)
    # without the input LayerNorm, plain projection
    refP, Pg = torch.zeros(M, N), torch.zeros(M, N, device="cuda")
    spec.rowtile_proj(X, M, D, Wqp, None, N, refP)
    hip.rowtile_proj(X.cuda(), M, D, Wqp.cuda(), None, N, Pg)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Pg.cpu().numpy(), refP.numpy(), atol=3e-4, rtol=3e-4)
    # output Linear + residual in place + LayerNorm of the result
    refX, refL = X.clone(), torch.full((M + 3, D), 2.0)
    spec.rowtile_proj(ATT, M, D, Wop, bo, D, refX, R=X, g2=g2, b2=b2, LN2=refL)
    Xg, Lg = X.cuda(), torch.full((M + 3, D), 2.0, device="cuda")
    hip.rowtile_proj(ATT.cuda(), M, D, Wop.cuda(), bo.cuda(), D, Xg, R=Xg, g2=g2.cuda(), b2=b2.cuda(), LN2=Lg)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Xg.cpu().numpy(), refX.numpy(), atol=3e-4, rtol=3e-4)
    np.testing.assert_allclose(Lg.cpu().numpy(), refL.numpy(), atol=5e-4, rtol=5e-4)


@pytest.mark.parametrize("M,D", [(42, 256), (2688, 256), (1000, 256), (333, 128)])
def test_rowtile_proj_fp16_weights(hip, M, D):
    """sc_rowtile_proj_h (v_mfma_f32_16x16x16_f16: fp16 weights, the normalised row tile rounded to fp16 when staged,
    fp32 accumulation / bias / residual / output LayerNorm) against fp32 arithmetic on the fp16-ROUNDED operands: what
    remains is the summation order (tolerance 2e-3 on values of magnitude 1)."""
    from speechcatcher_amd.weights import pack_panel_weight
    N = 3 * D
    X, ATT = _rand(M, D, seed=91), _rand(M, D, seed=92)
    Wqkv, bqkv = _rand(N, D, seed=93, scale=D ** -0.5), _rand(N, seed=94)
    Wo, bo = _rand(D, D, seed=95, scale=D ** -0.5), _rand(D, seed=96)
    g1, b1, g2, b2 = 1 + 0.1 * _rand(D, seed=97), _rand(D, seed=98), 1 + 0.1 * _rand(D, seed=99), _rand(D, seed=100)
    r16 = lambda t: t.half().float()   # noqa: E731
    ln = lambda t, g, b: torch.nn.functional.layer_norm(t, (D,), g, b, 1e-12)   # noqa: E731
    Wqh, Woh = pack_panel_weight(Wqkv).half(), pack_panel_weight(Wo).half()
    refQ = r16(ln(X, g1, b1)) @ r16(Wqkv).t() + bqkv
    Qg = torch.full((M + 3, N), 5.0, device="cuda")
    hip.rowtile_proj_h(X.cuda(), M, D, Wqh.cuda(), bqkv.cuda(), N, Qg, ln_g=g1.cuda(), ln_b=b1.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(Qg[:M].cpu().numpy(), refQ.numpy(), atol=2e-3, rtol=2e-3)
    assert float(Qg[M:].min()) == 5.0 and float(Qg[M:].max()) == 5.0          # rows >= M untouched
    refX = X + (r16(ATT) @ r16(Wo).t() + bo)
    Xg, Lg = X.cuda(), torch.full((M + 3, D), 2.0, device="cuda")
    hip.rowtile_proj_h(ATT.cuda(), M, D, Woh.cuda(), bo.cuda(), D, Xg, R=Xg, g2=g2.cuda(), b2=b2.cuda(), LN2=Lg)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Xg.cpu().numpy(), refX.numpy(), atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(Lg[:M].cpu().numpy(), ln(refX, g2, b2).numpy(), atol=3e-3, rtol=3e-3)


@pytest.mark.parametrize("M,D", [(10, 256), (210, 256), (2100, 256), (8400, 256), (77, 128)])
def test_rowtile_proj_split_weights(hip, M, D):
    """sc_rowtile_proj_s (fp16 hi | lo split of weights and row tile, three v_mfma_f32_16x16x32_f16 per product sum):
    against float64 as close as the fp32 kernel is - both forms of the kernel (q|k|v behind norm1; output Linear +
    residual + norm2), rows of very different magnitude."""
    from speechcatcher_amd.weights import pack_panel_weight, split_panel_weight
    N = 3 * D
    X, ATT = _rand(M, D, seed=91), _rand(M, D, seed=92)
    ATT *= torch.logspace(-3, 2, M)[:, None]
    Wqkv, bqkv = _rand(N, D, seed=93, scale=D ** -0.5), _rand(N, seed=94)
    Wo, bo = _rand(D, D, seed=95, scale=D ** -0.5), _rand(D, seed=96)
    g1, b1, g2, b2 = 1 + 0.1 * _rand(D, seed=97), _rand(D, seed=98), 1 + 0.1 * _rand(D, seed=99), _rand(D, seed=100)
    ln64 = lambda t, g, b: torch.nn.functional.layer_norm(t.double(), (D,), g.double(), b.double(), 1e-12)   # noqa: E731
    Wqp, Wop = pack_panel_weight(Wqkv), pack_panel_weight(Wo)
    Wqs, Wos = split_panel_weight(Wqp), split_panel_weight(Wop)
    xn = ln64(X, g1, b1)
    refQ = xn @ Wqkv.double().t() + bqkv.double()
    scaleQ = xn.abs() @ Wqkv.double().abs().t() + bqkv.double().abs()
    refX = X.double() + ATT.double() @ Wo.double().t() + bo.double()
    scaleX = ATT.double().abs() @ Wo.double().abs().t() + X.double().abs() + bo.double().abs()   # size of the summed terms
    err = {}
    for form in ("f32", "split"):
        Qg = torch.full((M + 3, N), 5.0, device="cuda")
        Xg, Lg = X.cuda(), torch.full((M + 3, D), 2.0, device="cuda")
        if form == "f32":
            hip.rowtile_proj(X.cuda(), M, D, Wqp.cuda(), bqkv.cuda(), N, Qg, ln_g=g1.cuda(), ln_b=b1.cuda())
            hip.rowtile_proj(ATT.cuda(), M, D, Wop.cuda(), bo.cuda(), D, Xg, R=Xg, g2=g2.cuda(), b2=b2.cuda(), LN2=Lg)
        else:
            hip.rowtile_proj_s(X.cuda(), M, D, Wqs.cuda(), bqkv.cuda(), N, Qg, ln_g=g1.cuda(), ln_b=b1.cuda())
            hip.rowtile_proj_s(ATT.cuda(), M, D, Wos.cuda(), bo.cuda(), D, Xg, R=Xg, g2=g2.cuda(), b2=b2.cuda(), LN2=Lg)
        torch.cuda.synchronize()
        assert float(Qg[M:].min()) == 5.0 and float(Qg[M:].max()) == 5.0 and float(Lg[M:].min()) == 2.0   # rows >= M untouched
        err[form] = (float(((Qg[:M].cpu().double() - refQ).abs() / scaleQ).max()),
                     float(((Xg.cpu().double() - refX).abs() / scaleX).max()))
        np.testing.assert_allclose(Lg[:M].cpu().numpy(), ln64(refX, g2, b2).float().numpy(), atol=2e-5, rtol=2e-5)
    for k in (0, 1):
        assert err["f32"][k] < 2e-6 and err["split"][k] < 2e-6, err
        assert err["split"][k] < 4.0 * err["f32"][k] + 2e-7, err


@pytest.mark.parametrize("M,D,F,N", [(10, 256, 2048, 768), (1280, 256, 2048, 768), (533, 256, 2048, 1024),
                                     (77, 128, 256, 384), (2100, 256, 2048, 768),
                                     (1280, 256, 2048, 1024), (1093, 256, 2048, 1024)])   # (two column blocks per workgroup)
def test_ffn_ln_proj_chain(hip, M, D, F, N):
    """Fused FFN whose reduce kernel also applies the LayerNorm and the next projection
    (x_in / x_out ping-pong, 4x4x1-MFMA row panels, N/D column blocks), with and without a row table."""
    from oracle.kernel_spec import SpecBackend
    from speechcatcher_amd.weights import pack_lane_weight, pack_panel_weight
    XN, X0 = _rand(M + 5, D, seed=71), _rand(M + 5, D, seed=72)
    W1, b1 = _rand(F, D, seed=73, scale=D ** -0.5), _rand(F, seed=74)
    W2, b2 = _rand(D, F, seed=75, scale=F ** -0.5), _rand(D, seed=76)
    Wq, bq = _rand(N, D, seed=77, scale=D ** -0.5), _rand(N, seed=78)
    g, be_ = 1 + 0.1 * _rand(D, seed=79), _rand(D, seed=80)
    W1p, W2p, Wqq = pack_panel_weight(W1), pack_panel_weight(W2), pack_lane_weight(Wq)
    spec = SpecBackend()
    for rows in (None, torch.randperm(M + 5, generator=torch.Generator().manual_seed(6))[:M].to(torch.int32)):
        refO, refQ = torch.full((M + 5, D), 7.0), torch.full((M + 5, N), 9.0)
        spec.ffn_ln_proj(XN, rows, M, D, F, W1p, b1, W2p, b2, X0, refO, g, be_, Wqq, bq, refQ, N)
        Og, Qg = torch.full((M + 5, D), 7.0, device="cuda"), torch.full((M + 5, N), 9.0, device="cuda")
        Xin = X0.cuda()
        hip.ffn_ln_proj(XN.cuda(), None if rows is None else rows.cuda(), M, D, F, W1p.cuda(), b1.cuda(), W2p.cuda(),
                        b2.cuda(), Xin, Og, g.cuda(), be_.cuda(), Wqq.cuda(), bq.cuda(), Qg, N)
        torch.cuda.synchronize()
        assert torch.equal(Xin.cpu(), X0)                        # x_in is read-only
        np.testing.assert_allclose(Og.cpu().numpy(), refO.numpy(), atol=3e-4, rtol=3e-4)   # untouched rows keep 7.0
        np.testing.assert_allclose(Qg.cpu().numpy(), refQ.numpy(), atol=6e-4, rtol=6e-4)


def test_row_compaction_tables(hip):
    """Ragged-batch compaction: the row-panel kernel and gemm_ln driven through a
    row table touch exactly the listed rows (others keep their content)."""
    from oracle.kernel_spec import SpecBackend
    from speechcatcher_amd.weights import pack_lane_weight as pack_panel_weight
    M, D, n = 200, 256, 53
    g0 = torch.Generator().manual_seed(77)
    rows = torch.randperm(M, generator=g0)[:n].to(torch.int32)
    A, W1, b1 = _rand(M, D, seed=41), _rand(D, D, seed=42, scale=D ** -0.5), _rand(D, seed=43)
    W2, b2 = _rand(D, D, seed=44, scale=D ** -0.5), _rand(D, seed=45)
    g, be_ = 1 + 0.1 * _rand(D, seed=46), _rand(D, seed=47)
    X0, N0, Q0 = _rand(M, D, seed=48), _rand(M, D, seed=49), _rand(M, D, seed=50)
    W1p, W2p = pack_panel_weight(W1), pack_panel_weight(W2)
    refX, refN, refQ = X0.clone(), N0.clone(), Q0.clone()
    spec = SpecBackend()
    spec.proj_ln_proj(A, D, W1p, b1, refX, D, g, be_, refN, W2p, b2, refQ, n, D, rows=rows)
    Xg, Ng, Qg = X0.cuda(), N0.cuda(), Q0.cuda()
    hip.proj_ln_proj(A.cuda(), D, W1p.cuda(), b1.cuda(), Xg, D, g.cuda(), be_.cuda(), Ng, W2p.cuda(), b2.cuda(), Qg,
                     n, D, rows=rows.cuda())
    torch.cuda.synchronize()
    for got, ref in ((Xg, refX), (Ng, refN), (Qg, refQ)):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=3e-4, rtol=3e-4)
    untouched = torch.ones(M, dtype=torch.bool)
    untouched[rows.long()] = False
    assert torch.equal(Xg.cpu()[untouched], X0[untouched]) and torch.equal(Qg.cpu()[untouched], Q0[untouched])
    # gemm_ln with the LayerNorm output following c_rows (split-K reduce+LN and the unfused path)
    for K in (256, 2048):
        A2, W = _rand(M, K, seed=51), _rand(D, K, seed=52, scale=K ** -0.5)
        refC, refL = X0.clone(), N0.clone()
        spec.gemm_ln(A2, rows, K, W, b1, refC, rows, D, n, D, K, g, be_, refL, residual=True, ln_at_crows=True)
        Cg, Lg = X0.cuda(), N0.cuda()
        hip.gemm_ln(A2.cuda(), rows.cuda(), K, W.cuda(), b1.cuda(), Cg, rows.cuda(), D, n, D, K, g.cuda(), be_.cuda(),
                    Lg, residual=True, ln_at_crows=True)
        torch.cuda.synchronize()
        np.testing.assert_allclose(Cg.cpu().numpy(), refC.numpy(), atol=3e-4, rtol=3e-4)
        np.testing.assert_allclose(Lg.cpu().numpy(), refL.numpy(), atol=3e-4, rtol=3e-4)


def test_decoder_without_row_panel_kernel(hip, monkeypatch):
    """SC_DEC_PANEL=0 keeps the GEMM / reduce+LN / GEMM form (also the path for
    feature dims the panel kernel does not cover): same trajectories."""
    monkeypatch.setenv("SC_DEC_PANEL", "0")
    from test_engine_spec import run_case
    run_case("tiny_c10240_b10_bbd0", backend=hip, device="cuda:0")


@pytest.mark.parametrize("d", [64, 256])
def test_layernorm(hip, d):
    M = 77
    x, g, b = _rand(M, d, seed=11, scale=3.0), 1 + 0.1 * _rand(d, seed=12), _rand(d, seed=13)
    x[5] = 0.0   # zero row: eps=1e-12 path
    ref = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-12)
    out = torch.zeros(M, d, device="cuda")
    hip.layernorm(x.cuda(), None, out, None, M, g.cuda(), b.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=2e-5)


def _lockstep_run(hip, case, n_calls=None, **kw):
    from lockstep import LockstepBackend
    from test_engine_spec import make_batch
    js, _ = load_case(case)
    meta = js["meta"]
    ls = LockstepBackend(hip, **kw)
    caps = dict(max_frames=256, max_tokens=200, pcm_capacity=1 << 18)
    sb_cpu = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], backend=ls, **caps)
    sb_gpu = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], backend=hip,
                        device="cuda:0", **caps)
    ls.attach(sb_cpu, sb_gpu)
    audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
    pos, chunk, k = 0, meta["chunk"], 0
    while pos < len(audio) and (n_calls is None or k < n_calls):
        end = min(pos + chunk, len(audio))
        sb_cpu.push([(0, audio[pos:end], end >= len(audio))])
        pos, k = end, k + 1
    return ls


def _dump(ls, name):
    import os
    rep = {"max_abs_diff": ls.report, "calls": ls.calls, "failures": ls.failures[:50],
           "int_mismatch": ls.int_mismatch[:50]}
    print(json.dumps(rep, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/{name}.json", "w") as f:
        json.dump(rep, f, indent=1)


ALL_OPS = {"logmel", "conv1", "gemm", "gemm_ln", "proj_ln_proj", "layernorm", "block_pack", "ctx_handoff", "enc_attention",
           "dec_self_attn", "dec_cross_attn", "logsoftmax_topk", "ctc_prefix_scan", "fuse_topw",
           "beam_prune", "ctc_gather_state", "ctc_extend_state", "dec_embed", "copy_rows",
           "log_softmax_rows"}


def test_every_kernel_lockstep_tiny(hip):
    """Whole tiny utterance, beam 10: each launched HIP kernel is compared with
    its spec on identical inputs (fp32 tolerance 2e-4 abs+rel; ints exact)."""
    ls = _lockstep_run(hip, "tiny_c10240_b10_bbd0")
    _dump(ls, "lockstep_tiny")
    assert not ls.failures, ls.failures[:10]
    assert not ls.int_mismatch, ls.int_mismatch[:10]
    assert set(ls.report) >= ALL_OPS


def test_every_kernel_lockstep_tiny_multiblock(hip):
    ls = _lockstep_run(hip, "tiny_c25600_b10_bbd0")
    assert not ls.failures, ls.failures[:10]
    assert not ls.int_mismatch, ls.int_mismatch[:10]


@pytest.mark.parametrize("layers", ["head_parallel", "head_parallel_hpw4", "stream_resident", "six_launch"])
def test_every_kernel_lockstep_xl(hip, monkeypatch, layers):
    """XL dims (d=256, 8 heads, 30+14 layers), first 5 calls of the fixture utterance; the forms of the decoder layers:
    3 launches per layer with one head per workgroup (small buckets), with four / two heads per workgroup (large
    buckets: 1024- / 512-thread workgroups, H/4 partial products per row) and the 6 launches of round 1."""
    from lockstep import LockstepBackend
    monkeypatch.setenv("SC_DEC_FUSED", "0" if layers == "six_launch" else "1")
    hpw = int(layers[-1]) if "hpw" in layers else 1
    monkeypatch.setenv("SC_DEC_HPW", str(hpw))
    monkeypatch.setattr(LockstepBackend, "dec_hpw", hpw)
    if hpw > 1:
        monkeypatch.setenv("SC_ATTN_DEEP", "0")      # (the few-streams variant runs one head per workgroup)
    monkeypatch.setattr(LockstepBackend, "fused_layers", layers != "six_launch")
    # (round 6) the stream-resident form: one workgroup per stream, both attentions of a layer in one launch
    monkeypatch.setenv("SC_DEC_STREAM", "1" if layers == "stream_resident" else "0")
    monkeypatch.setattr(LockstepBackend, "stream_layers", layers == "stream_resident")
    ls = _lockstep_run(hip, "xl_c10240_b10_bbd0", n_calls=4 if layers == "six_launch" else 5, atol=5e-4, rtol=5e-4)
    _dump(ls, "lockstep_xl_" + layers)
    assert ("dec_layer_self" in ls.calls) == (layers in ("head_parallel", "head_parallel_hpw4"))
    assert ("dec_layer_stream" in ls.calls) == (layers == "stream_resident")
    assert not ls.failures, ls.failures[:10]
    assert not ls.int_mismatch, ls.int_mismatch[:10]


@pytest.mark.parametrize("beam", [10, 5])
def test_every_kernel_lockstep_head_dim_64(hip, monkeypatch, beam):
    """the head-parallel layer kernels at the reference's DEFAULT geometry - 4 heads of 64 at d = 256, what
    speech2text_streaming.py:221-227, 236-244 builds when config.yaml names no heads (config.M_DEFAULTS with fewer layers).
    Until round 5 such a model took the six-launch decoder; since round 6 sc_dec_layer_fused_supported(256, 4, W, 2048) = 1
    (q|k|v projected in three passes of four column tiles, the output projection over the head's two k blocks).  Every op of
    the engine against its spec on identical inputs, beam 10 (10-row tiles) and beam 5 (5-row tiles)."""
    import test_engine_spec
    from lockstep import LockstepBackend
    from speechcatcher_amd.config import ModelConfig
    from speechcatcher_amd._abi import load
    from test_engine_spec import make_batch
    assert load().sc_dec_layer_fused_supported(256, 4, beam, 2048) == 1
    test_engine_spec.CFGS["M4"] = ModelConfig(d_model=256, enc_heads=4, enc_layers=3, dec_heads=4, dec_layers=2)
    monkeypatch.setenv("SC_DEC_FUSED", "1")
    monkeypatch.setattr(LockstepBackend, "fused_layers", True)
    ls = LockstepBackend(hip, atol=5e-4, rtol=5e-4)
    caps = dict(max_frames=256, max_tokens=200, pcm_capacity=1 << 18)
    sb_cpu = make_batch("M4", 1234, "meanstd", beam, False, backend=ls, **caps)
    sb_gpu = make_batch("M4", 1234, "meanstd", beam, False, backend=hip, device="cuda:0", **caps)
    assert sb_cpu.ph1 is not None
    ls.attach(sb_cpu, sb_gpu)
    audio = synth.synth_audio(77, 10240 * 6)
    for k in range(6):
        sb_cpu.push([(0, audio[k * 10240:(k + 1) * 10240], False)])
    _dump(ls, f"lockstep_head_dim_64_beam{beam}")
    assert ls.calls.get("dec_layer_self", 0) > 0 and ls.calls.get("dec_layer_cross", 0) > 0
    assert not ls.failures, ls.failures[:10]
    assert not ls.int_mismatch, ls.int_mismatch[:10]


def test_encoder_layers_rowtile(hip, monkeypatch):
    """sc_encoder_layers with the row-tile projections (norm1 + q|k|v, output Linear + residual + norm2 in one
    launch each) and with the LayerNorm + GEMM launches they replace, both against the spec encoder."""
    from oracle.kernel_spec import SpecBackend
    from test_engine_spec import make_batch
    js, _ = load_case("xl_c10240_b10_bbd0")
    meta = js["meta"]
    caps = dict(max_frames=64, max_tokens=20, pcm_capacity=1 << 16)
    spec = SpecBackend()
    sb_cpu = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], backend=spec, **caps)
    sb_gpu = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], backend=hip,
                        device="cuda:0", **caps)
    cfg = sb_cpu.cfg
    # 6 of the 30 layers and inputs at the scale of x*sqrt(d) + PE: deep stacks of random layers amplify
    # rounding differences of ANY two implementations (the end-to-end cases pin the full depth)
    sb_cpu.w.enc, sb_gpu.w.enc = sb_cpu.w.enc[:6], sb_gpu.w.enc[:6]
    d, F, nl = cfg.d_model, cfg.ffn_dim, 6
    ns, nbk, R = 4, 4, 42
    nblk, M = ns * nbk, ns * nbk * 42
    x0 = 8.0 * _rand(M, d, seed=301)
    jobs = torch.tensor([[s * nbk, nbk, s * nl, s % 2] for s in range(ns)], dtype=torch.int32)
    ctx0 = _rand(ns * nl, d, seed=302)

    def run(be, w, dev):
        x, ctx = x0.clone().to(dev), ctx0.clone().to(dev)   # .to('cpu') alone would alias the inputs
        xn, att = torch.zeros(M, d, device=dev), torch.zeros(M, d, device=dev)
        qkv, ffh = torch.zeros(M, 3 * d, device=dev), torch.zeros(M, F, device=dev)
        be.encoder_layers(w, x, nblk, R, True, jobs.to(dev), ns, ctx, xn, qkv, att, ffh)
        if dev != "cpu":
            torch.cuda.synchronize()
        return x.cpu(), ctx.cpu()

    ref_x, ref_ctx = run(spec, sb_cpu.w, "cpu")
    graphs, hip.use_graphs = hip.use_graphs, False   # the switch is read when the launches are issued
    try:
        outs = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("SC_ENC_ROWTILE", mode)
            outs[mode] = run(hip, sb_gpu.w, "cuda:0")
    finally:
        hip.use_graphs = graphs
    scale = float(ref_x.abs().max())
    for mode, (gx, gctx) in outs.items():
        assert float((gx - ref_x).abs().max()) <= 1e-3 * max(scale, 1.0), mode
        assert float((gctx - ref_ctx).abs().max()) <= 1e-3 * max(scale, 1.0), mode


@pytest.mark.parametrize("dims", ["xl", "tiny"])
def test_attention_full_batch_variants_lockstep(hip, monkeypatch, dims):
    """The attention kernels of FULL batches (2 key tiles per wave in flight, <= 128 VGPRs) are picked by batch
    size; force them on the single-stream fixtures so that every launch is compared with its spec on identical
    inputs: XL dims (d_k = 32, head-parallel layer kernels), tiny dims (d_k = 16, stand-alone attention kernels,
    whole utterance)."""
    from lockstep import LockstepBackend
    monkeypatch.setenv("SC_ATTN_DEEP", "0")
    if dims == "tiny":
        ls = _lockstep_run(hip, "tiny_c10240_b10_bbd0")
        assert ls.calls.get("dec_self_attn", 0) > 0 and ls.calls.get("dec_cross_attn", 0) > 0
    else:   # first decode blocks
        ls = _lockstep_run(hip, "xl_c10240_b10_bbd0", n_calls=5, atol=5e-4, rtol=5e-4)
        assert ls.calls.get("dec_layer_self", 0) > 0 and ls.calls.get("dec_layer_cross", 0) > 0
    assert not ls.failures, ls.failures[:10]
    assert not ls.int_mismatch, ls.int_mismatch[:10]


def _scan_setup(hip, S, T, L, has, seed=0):
    """Two identical batches (CPU spec / GPU) with a random CTC table of T frames, random pre-beam candidates
    (incl. the last token, eos and blank) and, if `has`, random forward variables of the previous prefix."""
    from oracle.kernel_spec import SpecBackend
    from test_engine_spec import make_batch
    # (max_frames >= 512: the T-parallel scan parks 32 segment states per pair where the 16-frame checkpoints live and is
    # switched off for tables shorter than that)
    kw = dict(n_streams=S, max_frames=max(T + 12, 512 if T >= 64 else 0), max_tokens=L + 8, pcm_capacity=1 << 12)
    sc = make_batch("TINY", 1234, "meanstd", 10, False, backend=SpecBackend(), **kw)
    sg = make_batch("TINY", 1234, "meanstd", 10, False, backend=hip, device="cuda:0", **kw)
    g = torch.Generator().manual_seed(seed)
    V, W, K = sc.cfg.vocab_size, sc.W, sc.K
    x = torch.log_softmax(torch.randn(S, sc.TCAP, V, generator=g) * 2.0, -1)
    x[:, 24:] = torch.randn(S, sc.TCAP - 24, V, generator=g) * 2.0 - 4.0      # later rows are raw logits (A1)
    sc.ctcx.view(S, sc.TCAP, V).copy_(x)
    sc.ctcxT.view(S, V, -1)[:, :, :sc.TCAP] = x.transpose(1, 2)
    for s in range(S):
        for h in range(W):
            ids = torch.randperm(V - 2, generator=g)[:K] + 1
            last = int(ids[3])
            ids[5], ids[7] = V - 1, 0                                        # eos and blank among the candidates
            sc.pre_ids[s * W + h] = ids.to(torch.int32)
            sc.yseq[0, s, h, :L] = torch.randint(1, V - 1, (L,), generator=g).to(torch.int32)
            sc.yseq[0, s, h, L - 1] = last
    if has:
        r = torch.randn(S, sc.TCAP, 2, W, generator=g) * 3.0 - 30.0
        sc.ctc_r[0].copy_(r)
        sc.ctc_rs[0].copy_(torch.logsumexp(r, 2))        # (scasr.h: sc_search.ctc_rs is kept wherever ctc_r is written)
    sc.ctrl.copy_(torch.tensor([[1, 0, 0, T, L, W, int(has), 0]] * S, dtype=torch.int32))
    for k, v in vars(sc).items():
        if isinstance(v, torch.Tensor) and k in ("ctcx", "ctcxT", "pre_ids", "yseq", "ctc_r", "ctc_rs", "ctrl"):
            getattr(sg, k).copy_(v)
    return sc, sg


@pytest.mark.parametrize("split", [0, 48])
@pytest.mark.parametrize("T,L,has", [(4500, 300, True), (4500, 1, False), (37, 9, True), (16, 2, False), (450, 120, True),
                                     (300, 1, False), (1000, 960, True)])
def test_ctc_prefix_scan_long_table(hip, T, L, has, split):
    """CTCPrefixScoreTH.__call__ (ctc_prefix_score_full.py:146-291) at the table length of a 180 s CLI segment
    (T = 4500 encoder frames) and at chunk-boundary lengths: the column-streaming HIP scan against its spec -
    log psi of all W x K candidates, the eos score and every forward variable r[t]; split = 48: tables with at least
    48 frames to walk take the T-parallel kernel (16 segments per pair), the short ones stay sequential."""
    from oracle.kernel_spec import SpecBackend
    sc, sg = _scan_setup(hip, 2, T, L, has)
    SpecBackend().ctc_prefix_scan(sc)
    hip.ctc_prefix_scan(sg, split_min=split)
    torch.cuda.synchronize()
    W, K = sc.W, sc.K
    psi_c, psi_g = sc.psi.view(-1), sg.psi.cpu().view(-1)
    fin = psi_c > -1e9
    assert torch.equal(fin, psi_g > -1e9)
    tol = 2e-4 + 2e-6 * T      # T dependent log-add-exps in fp32 on values of magnitude 1e1..1e4
    assert float((psi_c[fin] - psi_g[fin]).abs().max()) <= tol * max(1.0, float(psi_c[fin].abs().max()) * 1e-2)
    np.testing.assert_allclose(sg.psi_eos.cpu().numpy(), sc.psi_eos.numpy(), rtol=1e-5, atol=1e-3)
    # the forward variables: W winners (hypothesis h, candidate k) through ctc_gather_state - the GPU rebuilds their r[t] at
    # every frame from what its scan left (16-frame checkpoints, or the segment states of the T-parallel form: round 5)
    g = torch.Generator().manual_seed(7)
    for s_ in range(sc.S):
        sel = torch.stack([torch.randint(0, W, (W,), generator=g), torch.randint(0, K, (W,), generator=g)], 1).to(torch.int32)
        sc.sel[s_].copy_(sel)
        sg.sel[s_].copy_(sel)
    SpecBackend().ctc_gather_state(sc)
    hip.ctc_gather_state(sg, split_min=split)
    torch.cuda.synchronize()
    rc = sc.ctc_r[1, :, :T]
    rg = sg.ctc_r[1, :, :T].cpu()
    live = rc > -1e9
    assert torch.equal(live, rg > -1e9)
    assert float(((rc - rg).abs() / rc.abs().clamp(min=1.0))[live].max()) <= 1e-4
    if split == 0:                                      # checkpoints of the sequential scan: r at the frames t % 16 == 15
        nck = T // 16
        rc = sc.ctc_rnew.view(2, -1, 2, W * K)[:, :nck]
        rg = sg.ctc_rnew.cpu().view(2, -1, 2, W * K)[:, :nck]
        live = rc > -1e9
        assert torch.equal(live, rg > -1e9)
        assert float(((rc - rg).abs() / rc.abs().clamp(min=1.0))[live].max()) <= 1e-4


def test_split_scan_followed_by_the_legacy_state_rebuild_is_refused(hip):
    """since the T-parallel scan parks SEGMENT START states where the sequential scan leaves its 16-frame
    checkpoints, sc_ctc_gather_state() (split_min = 0) behind sc_ctc_prefix_scan_split(sb, n > 0) rebuilt the CTC state from
    the wrong kind of rows without an error.  The library remembers the effective split_min of a batch's last scan and
    refuses a rebuild that names another one; the matching value and the un-split pair still work.
    Reference: the state the next step's scan starts from (ctc_prefix_score_full.py:293-330 select_state)."""
    from speechcatcher_amd._abi import ScasrError
    sc, sg = _scan_setup(hip, 2, 450, 120, True)
    W, K = sc.W, sc.K
    g = torch.Generator().manual_seed(11)
    for s_ in range(sc.S):
        sg.sel[s_].copy_(torch.stack([torch.randint(0, W, (W,), generator=g), torch.randint(0, K, (W,), generator=g)], 1).to(torch.int32))
    hip.ctc_prefix_scan(sg, split_min=48)
    with pytest.raises(ScasrError, match="split_min"):
        hip.ctc_gather_state(sg)                 # the legacy entry point: split_min = 0
    hip.ctc_gather_state(sg, split_min=48)       # the scan's own value
    hip.ctc_prefix_scan(sg)
    with pytest.raises(ScasrError, match="split_min"):
        hip.ctc_gather_state(sg, split_min=48)
    hip.ctc_gather_state(sg)
    torch.cuda.synchronize()


def test_ctc_prefix_scan_time_at_long_tables(hip):
    """Time per launch of the scan at T = 4500 (8 active streams, hypotheses of 300 tokens): the column-streaming
    kernel and its T-parallel form; numbers -> gpurun_out/r03_ctc_scan_timing.json (round 2 also timed the row-gather
    kernel they replaced: 1233 us at T = 4500, docs/profiles_r1-r3/r02_ctc_scan_timing.json)."""
    import json
    import os
    out = {}
    for T, L in ((450, 250), (450, 60), (4500, 300)):
        sc, sg = _scan_setup(hip, 8, T, L, True)
        st = hip.search_struct(sg)
        for name in ("column_streaming", "t_parallel"):
            split = 256 if name == "t_parallel" else 0
            for _ in range(3):
                hip.ctc_prefix_scan(sg, split_min=split)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                hip.ctc_prefix_scan(sg, split_min=split)
            b.record()
            torch.cuda.synchronize()
            out[f"T{T}{'_L60' if L == 60 else ''}_{name}_us"] = round(a.elapsed_time(b) * 1e3 / 20, 1)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r03_ctc_scan_timing.json", "w") as f:
        json.dump(out, f)
    print(out)
    # 4200 sequential frames x ~45 dependent-ish instructions of ONE wave per SIMD: ~0.16 us per frame measured
    # (r01: 0.29); the next step is splitting the r^n / r^b / psi chains over waves of different SIMDs
    assert out["T4500_column_streaming_us"] < 800.0
    assert out["T4500_t_parallel_us"] < 150.0
    # (T = 450 with 250-token hypotheses has 201 frames to walk: below the 256-frame threshold, not split)
