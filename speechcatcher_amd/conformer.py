"""Conformer building blocks on the HIP kernels:
ConvolutionModule and RelPositionMultiHeadedAttention are named by the north
star but no shipped speechcatcher model uses them (the encoders are
contextual-block *Transformers*), so they are provided as stand-alone ops with
module-level parity against the reference classes."""
import math

import torch


def pack_conv_module(sd, device):
    """reference parameter names -> GEMM-friendly tensors (Conv1d k=1 == Linear)."""
    dev = lambda t: t.detach().float().contiguous().to(device)  # noqa: E731
    C = sd["layernorm.weight"].numel()
    return {
        "C": C, "k": sd["depthwise_conv.weight"].shape[-1],
        "ln_g": dev(sd["layernorm.weight"]), "ln_b": dev(sd["layernorm.bias"]),
        "pw1_w": dev(sd["pointwise_conv1.weight"].reshape(2 * C, C)), "pw1_b": dev(sd["pointwise_conv1.bias"]),
        "dw_w": dev(sd["depthwise_conv.weight"].reshape(C, -1)), "dw_b": dev(sd["depthwise_conv.bias"]),
        "bn_g": dev(sd["batch_norm.weight"]), "bn_b": dev(sd["batch_norm.bias"]),
        "bn_mean": dev(sd["batch_norm.running_mean"]), "bn_var": dev(sd["batch_norm.running_var"]),
        "pw2_w": dev(sd["pointwise_conv2.weight"].reshape(C, C)), "pw2_b": dev(sd["pointwise_conv2.bias"]),
    }


def conv_module(be, w, x):
    """ConvolutionModule.forward (model/layers/convolution.py:84-120), eval mode.
    x (B, T, C) on the backend's device -> (B, T, C)."""
    B, T, C = x.shape
    M = B * T
    x2 = x.reshape(M, C).contiguous()
    xn = torch.empty_like(x2)
    be.layernorm(x2, None, xn, None, M, w["ln_g"], w["ln_b"], eps=1e-5)   # nn.LayerNorm default eps
    y = torch.empty(M, 2 * C, device=x.device)
    be.gemm(xn, None, C, w["pw1_w"], w["pw1_b"], y, None, 2 * C, M, 2 * C, C)
    z = torch.empty(M, C, device=x.device)
    be.glu_dwconv_bn_swish(y, B, T, C, w["k"], w["dw_w"], w["dw_b"], w["bn_g"], w["bn_b"], w["bn_mean"],
                           w["bn_var"], 1e-5, z)
    out = torch.empty(M, C, device=x.device)
    be.gemm(z, None, C, w["pw2_w"], w["pw2_b"], out, None, C, M, C, C)
    return out.view(B, T, C)


def pack_relpos_mha(sd, device):
    dev = lambda t: t.detach().float().contiguous().to(device)  # noqa: E731
    return {
        "wqkv": dev(torch.cat([sd["linear_q.weight"], sd["linear_k.weight"], sd["linear_v.weight"]], 0)),
        "bqkv": dev(torch.cat([sd["linear_q.bias"], sd["linear_k.bias"], sd["linear_v.bias"]], 0)),
        "wpos": dev(sd["linear_pos.weight"]), "bias_u": dev(sd["pos_bias_u"]), "bias_v": dev(sd["pos_bias_v"]),
        "wo": dev(sd["linear_out.weight"]), "bo": dev(sd["linear_out.bias"]),
        "H": sd["pos_bias_u"].shape[0],
    }


def relpos_mha(be, w, x, pos_emb, mask=None):
    """RelPositionMultiHeadedAttention.forward(x, x, x, pos_emb, mask)
    (model/attention/multi_head_attention.py:316-378).  x (B, T, d), pos_emb (T, d); mask None, (B, 1, T) or
    (B, T, T) like the reference's (0 = masked out); any T."""
    if mask is not None:
        mask = (mask != 0).to(torch.uint8)
        mask = (mask.reshape(mask.shape[0], -1) if mask.shape[1] == 1 else mask).contiguous().to(x.device)
    B, T, d = x.shape
    M = B * T
    x2 = x.reshape(M, d).contiguous()
    qkv = torch.empty(M, 3 * d, device=x.device)
    be.gemm(x2, None, d, w["wqkv"], w["bqkv"], qkv, None, 3 * d, M, 3 * d, d)
    p = torch.empty(T, d, device=x.device)
    be.gemm(pos_emb.contiguous(), None, d, w["wpos"], None, p, None, d, T, d, d)
    ctx = torch.empty(M, d, device=x.device)
    be.relpos_attention(qkv, p, w["bias_u"], w["bias_v"], ctx, B, T, w["H"], mask=mask)
    out = torch.empty(M, d, device=x.device)
    be.gemm(ctx, None, d, w["wo"], w["bo"], out, None, d, M, d, d)
    return out.view(B, T, d)


def rel_positional_encoding(pe_table, x, offset=0):
    """RelPositionalEncoding.forward (model/layers/positional_encoding.py:97-122):
    (x*sqrt(d) + pe, pe).  Same arithmetic as the block_pack kernel's frame rows."""
    d = x.shape[-1]
    pe = pe_table[offset: offset + x.shape[1]]
    return x * math.sqrt(d) + pe, pe
