"""ESPnet state dict -> packed fp32 tensors laid out for the HIP kernels.

PyTorch is used here only to load / hold weight tensors (north_star:
"PyTorch-ROCm only for checkpoint load/weight tensors").  Layout choices
(see DESIGN.md "Data layout in HBM"):

* every Linear keeps its [out, in] row-major weight: the GEMM kernel computes
  C = A . W^T with both operands K-contiguous;
* Q/K/V projections are concatenated to one [3d, d] matrix (encoder self-attn,
  decoder self-attn) and K/V to [2d, d] (decoder src-attn);
* Conv2d #2 weight [co, ci, kh, kw] is permuted to [co, (kh, kw, ci)] so the
  implicit-GEMM K axis walks contiguous channels of the channels-last conv1
  output;
* the subsampling output Linear [d, c*F2 + f] is permuted to [d, f*d + c] to
  consume conv2's (t, f, c) output without a transpose.

reference schema: SURVEY.md Appendix B /
speechcatcher/model/checkpoint_loader.py:134-149.
"""
from typing import Dict

import numpy as np
import torch

from .config import ModelConfig
from .mel import (fft_twiddles, hann_window_periodic, melscale_fbanks_slaney,
                  positional_encoding_table)


PANEL_DIMS = (64, 128, 256)   # feature dims of the row-panel kernel (sc_proj_ln_proj_supported)


def pack_panel_weight(W: torch.Tensor) -> torch.Tensor:
    """[N][K] Linear weight -> MFMA fragment order of sc_proj_ln_proj / sc_ffn_ln
    (include/scasr.h: out[tile][ki][half][lane = kk*16 + r][c] =
    W[tile*16 + r][ki*32 + 8*kk + 4*half + c]); a pure permutation."""
    N, K = W.shape
    assert N % 16 == 0 and K % 32 == 0
    return W.reshape(N // 16, 16, K // 32, 4, 2, 4).permute(0, 2, 4, 3, 1, 5).contiguous().reshape(N, K)


def unpack_panel_weight(Wp: torch.Tensor) -> torch.Tensor:
    N, K = Wp.shape
    return Wp.reshape(N // 16, K // 32, 2, 4, 16, 4).permute(0, 4, 1, 3, 2, 5).contiguous().reshape(N, K)


def split_panel_weight(Wp: torch.Tensor) -> torch.Tensor:
    """fp16 hi | lo split of a ``pack_panel_weight`` copy for the split-precision fused feed-forward (sc_ffn_ln_s,
    csrc/gemm.hip ffn_fused_kernel WF = 2): per (tile, ki) the first 64-lane slab holds, per lane, the fp16 roundings
    ``hi`` of its 8 k values (those of both fp32 slabs), the second slab ``fp16((w - hi) * 2**11)`` - the same bytes
    as the fp32 copy.  w = hi + lo / 2**11 up to ~2**-23 |w|."""
    N, K = Wp.shape
    v = Wp.reshape(-1, 2, 64, 4)
    v = torch.cat([v[:, 0], v[:, 1]], dim=-1)                 # [n][lane][8 k values]
    assert float(v.abs().max()) < 65504.0, "weights beyond fp16's range cannot be split"
    hi = v.to(torch.float16)
    lo = ((v - hi.to(torch.float32)) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo], dim=1).contiguous().reshape(N, 2 * K)


def split16_operand_bounds(lw: Dict[str, torch.Tensor], ln: str, d: int) -> Dict[str, float]:
    """Upper bounds of the ACTIVATIONS the split-precision kernels split into fp16 hi | lo (ffn_fused_kernel /
    rowtile_proj_kernel, WF = 2: "operands must lie within fp16's range"), from the weights alone: a LayerNorm output is at
    most sqrt(d-1)*|g_i| + |b_i| in element i (the largest z-score d values can hold), a Linear's output at most
    |W| @ that + |bias|.  ``ln``: the LayerNorm in front of the feed-forward ("ln2" encoder, "ln3" decoder)."""
    z = float(np.sqrt(d - 1.0))
    out = {}
    bx = z * lw[ln + "_g"].abs() + lw[ln + "_b"].abs()
    out["ffn_in"] = float(bx.max())
    out["ffn_hidden"] = float((lw["w1"].abs() @ bx + lw["b1"].abs()).max())
    if ln == "ln2":   # encoder layer: Q|K|V projection behind norm1, output projection behind the attention (a convex
        b1 = z * lw["ln1_g"].abs() + lw["ln1_b"].abs()   # combination of V rows)
        out["proj_in"] = float(b1.max())
        out["attn_context"] = float((lw["wqkv"][2 * d:].abs() @ b1 + lw["bqkv"][2 * d:].abs()).max())
    return out


def pack_lane_weight(W: torch.Tensor) -> torch.Tensor:
    """[N][K] Linear weight -> lane order of sc_proj_ln_proj (include/scasr.h:
    out[tile][q][lane][c] = W[tile*64 + lane][4*q + c]); a pure permutation."""
    N, K = W.shape
    assert N % 64 == 0 and K % 4 == 0
    return W.reshape(N // 64, 64, K // 4, 4).permute(0, 2, 1, 3).contiguous().reshape(N, K)


def unpack_lane_weight(Wq: torch.Tensor) -> torch.Tensor:
    N, K = Wq.shape
    return Wq.reshape(N // 64, K // 4, 64, 4).permute(0, 2, 1, 3).contiguous().reshape(N, K)


def ffn_fused_supported(d: int, F: int) -> bool:
    """sc_ffn_ln_supported"""
    return d in (128, 256) and F % 128 == 0 and F >= 128


def dec_layer_fused_supported(cfg, W: int) -> bool:
    """sc_dec_layer_fused_supported + the output layer condition of sc_decode_step (V a multiple of d)"""
    d, H = cfg.d_model, cfg.dec_heads
    return (d in (128, 256) and d % H == 0 and (d // H in (16, 32) or (d == 256 and d // H == 64)) and 1 <= W <= 16
            and ffn_fused_supported(d, cfg.ffn_dim) and cfg.vocab_size % d == 0)


def rowtile_proj_supported(d: int, N: int) -> bool:
    """sc_rowtile_proj_supported"""
    return d in (128, 256) and N % 128 == 0 and N >= 128


class PackedWeights:
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: ModelConfig, device,
                 mean=None, std=None, ffn_dtype: str = "float32", proj_dtype: str = "float32",
                 dec_dtype: str = "float32"):
        """``ffn_dtype="float16"``: additional fp16 copies of the fragment-packed feed-forward weights (w1_h / w2_h);
        the fused FFN kernels then run fp16 MFMA inputs with fp32 accumulation (BASELINE configs[4]; never the
        default - the reference computes in fp32).  ``proj_dtype="float16"``: the same for the attention projections
        of the encoder layers (wqkv_h / wo_h: sc_rowtile_proj_h; the decoder's row panels stay fp32).
        ``"split16"`` (either): fp16 hi | lo splits of the fp32 weights instead (``split_panel_weight``: w1_s / w2_s,
        wqkv_s / wo_s) - the kernels split the activations the same way and evaluate every product sum with three fp16
        MFMAs: fp32-grade results (all reference fixtures at the fp32 tolerance) at a fraction of the f32 matrix-pipe
        time.  Opt-in as well: the default computes on the f32 MFMA path.
        ``dec_dtype="float16"`` (round 4, the decoder side of BASELINE configs[4]): fp16 copies of the decoder's attention
        projections (wqkv_pph / wq_pph / wo_pph / wo2_pph: sc_dec_layer_self / _cross) - fp16 MFMA inputs there - and the
        partial products between the decoder's kernels are stored in fp16 (sc_search.act_half).  The fp16 copy of the
        output layer (out_w_qh) is carried too but NOT used by the engines: logits feed the scores directly and fp16 inputs
        there moved the best hypothesis of 18 of 256 test streams (the rest of the fp16 decoder mode: 1).  LayerNorm,
        softmax, log-softmax, the output layer, the CTC scan and all scores stay fp32."""
        assert ffn_dtype in ("float32", "float16", "split16") and proj_dtype in ("float32", "float16", "split16")
        assert dec_dtype in ("float32", "float16")
        self.ffn_dtype, self.proj_dtype, self.dec_dtype = ffn_dtype, proj_dtype, dec_dtype
        self.cfg = cfg
        self.device = torch.device(device)
        d, F2 = cfg.d_model, cfg.conv_freq2

        def dev(t):
            return t.detach().to(torch.float32).contiguous().to(self.device)

        g = lambda k: sd[k].detach().to(torch.float32)  # noqa: E731
        # --- frontend tables
        self.window = dev(hann_window_periodic(cfg.win_length))
        self.mel_fb = dev(melscale_fbanks_slaney(cfg.n_fft // 2 + 1, 0.0, cfg.sample_rate / 2.0,
                                                 cfg.n_mels, cfg.sample_rate))
        self.twiddle = dev(torch.from_numpy(fft_twiddles(cfg.n_fft)))
        self.pe = dev(positional_encoding_table(cfg.pe_max_len, d))
        # MVN in float64, as numpy does in the reference (A11)
        if mean is None or std is None:
            self.has_mvn = False
            mean = np.zeros(cfg.n_mels)
            std = np.ones(cfg.n_mels)
        else:
            self.has_mvn = True
        self.mvn_is_f64 = (np.asarray(mean).dtype == np.float64)
        self.mean64 = torch.from_numpy(np.asarray(mean, dtype=np.float64)).to(self.device)
        self.std64 = torch.from_numpy(np.asarray(std, dtype=np.float64)).to(self.device)
        # --- subsampling
        self.conv1_w = dev(g("encoder.embed.conv.0.weight").reshape(d, 9))
        self.conv1_b = dev(g("encoder.embed.conv.0.bias"))
        w2 = g("encoder.embed.conv.2.weight")  # [co, ci, kh, kw]
        self.conv2_w = dev(w2.permute(0, 2, 3, 1).reshape(d, 9 * d))
        self.conv2_b = dev(g("encoder.embed.conv.2.bias"))
        wo = g("encoder.embed.out.weight")  # [d, c*F2 + f]
        self.sub_out_w = dev(wo.view(d, d, F2).permute(0, 2, 1).reshape(d, F2 * d))
        self.sub_out_b = dev(g("encoder.embed.out.bias"))
        # --- encoder layers
        self.enc = []
        for i in range(cfg.enc_layers):
            p = f"encoder.encoders.{i}"
            a = p + ".self_attn"
            self.enc.append(dict(
                ln1_g=dev(g(p + ".norm1.weight")), ln1_b=dev(g(p + ".norm1.bias")),
                wqkv=dev(torch.cat([g(a + ".linear_q.weight"), g(a + ".linear_k.weight"), g(a + ".linear_v.weight")], 0)),
                bqkv=dev(torch.cat([g(a + ".linear_q.bias"), g(a + ".linear_k.bias"), g(a + ".linear_v.bias")], 0)),
                wo=dev(g(a + ".linear_out.weight")), bo=dev(g(a + ".linear_out.bias")),
                ln2_g=dev(g(p + ".norm2.weight")), ln2_b=dev(g(p + ".norm2.bias")),
                w1=dev(g(p + ".feed_forward.w_1.weight")), b1=dev(g(p + ".feed_forward.w_1.bias")),
                w2=dev(g(p + ".feed_forward.w_2.weight")), b2=dev(g(p + ".feed_forward.w_2.bias")),
            ))
        self.enc_norm_g = dev(g("encoder.after_norm.weight"))
        self.enc_norm_b = dev(g("encoder.after_norm.bias"))
        # --- decoder
        self.embed = dev(g("decoder.embed.0.weight"))
        self.dec = []
        for i in range(cfg.dec_layers):
            p = f"decoder.decoders.{i}"
            a, c = p + ".self_attn", p + ".src_attn"
            self.dec.append(dict(
                ln1_g=dev(g(p + ".norm1.weight")), ln1_b=dev(g(p + ".norm1.bias")),
                wqkv=dev(torch.cat([g(a + ".linear_q.weight"), g(a + ".linear_k.weight"), g(a + ".linear_v.weight")], 0)),
                bqkv=dev(torch.cat([g(a + ".linear_q.bias"), g(a + ".linear_k.bias"), g(a + ".linear_v.bias")], 0)),
                wo=dev(g(a + ".linear_out.weight")), bo=dev(g(a + ".linear_out.bias")),
                ln2_g=dev(g(p + ".norm2.weight")), ln2_b=dev(g(p + ".norm2.bias")),
                wq=dev(g(c + ".linear_q.weight")), bq=dev(g(c + ".linear_q.bias")),
                wkv=dev(torch.cat([g(c + ".linear_k.weight"), g(c + ".linear_v.weight")], 0)),
                bkv=dev(torch.cat([g(c + ".linear_k.bias"), g(c + ".linear_v.bias")], 0)),
                wo2=dev(g(c + ".linear_out.weight")), bo2=dev(g(c + ".linear_out.bias")),
                ln3_g=dev(g(p + ".norm3.weight")), ln3_b=dev(g(p + ".norm3.bias")),
                w1=dev(g(p + ".feed_forward.w_1.weight")), b1=dev(g(p + ".feed_forward.w_1.bias")),
                w2=dev(g(p + ".feed_forward.w_2.weight")), b2=dev(g(p + ".feed_forward.w_2.bias")),
            ))
        for lw in self.dec:   # fragment-ordered copies for the row-panel kernel
            for n in ("wo", "wq", "wo2"):
                lw[n + "_p"] = pack_lane_weight(lw[n]) if d in PANEL_DIMS else lw[n]
        for lw in self.dec:              # Q|K|V in lane order: projected by the reduce kernel of the layer before
            lw["wqkv_q"] = pack_lane_weight(lw["wqkv"]) if d in PANEL_DIMS else lw["wqkv"]
        for lw in self.dec:              # Q|K|V and the cross-attention query projection in MFMA-fragment order:
            if d in (128, 256):          # head-parallel decoder layers (sc_dec_layer_self / _cross)
                lw["wqkv_pp"] = pack_panel_weight(lw["wqkv"])
                lw["wq_pp"] = pack_panel_weight(lw["wq"])
                lw["wo_pp"] = pack_panel_weight(lw["wo"])
                lw["wo2_pp"] = pack_panel_weight(lw["wo2"])
                if dec_dtype == "float16":   # same fragment order, 2-byte elements: fp16 MFMA inputs in the layer kernels
                    for n in ("wqkv", "wq", "wo", "wo2"):
                        lw[n + "_pph"] = lw[n + "_pp"].to(torch.float16).contiguous()
        for lw in self.enc:              # encoder attention projections for the row-tile kernel (sc_rowtile_proj)
            for n in ("wqkv", "wo"):
                lw[n + "_p"] = pack_panel_weight(lw[n]) if rowtile_proj_supported(d, d) else lw[n]
                if proj_dtype == "float16" and rowtile_proj_supported(d, d):
                    lw[n + "_h"] = lw[n + "_p"].to(torch.float16).contiguous()   # same fragment order, 2-byte elements
                if proj_dtype == "split16" and rowtile_proj_supported(d, d):
                    lw[n + "_s"] = split_panel_weight(lw[n + "_p"])
        for lw in self.enc + self.dec:   # ... and for the fused feed-forward kernel
            for n in ("w1", "w2"):
                lw[n + "_p"] = pack_panel_weight(lw[n]) if ffn_fused_supported(d, cfg.ffn_dim) else lw[n]
                if ffn_dtype == "float16" and ffn_fused_supported(d, cfg.ffn_dim):
                    lw[n + "_h"] = lw[n + "_p"].to(torch.float16).contiguous()   # same fragment order, 2-byte elements
                if ffn_dtype == "split16" and ffn_fused_supported(d, cfg.ffn_dim):
                    lw[n + "_s"] = split_panel_weight(lw[n + "_p"])
        # split16: no activation the kernels split may leave fp16's range (it would become inf silently) - decided here,
        # from the weights, for every layer that takes the split path; a model that fails it must use float32 / float16
        if "split16" in (ffn_dtype, proj_dtype):
            for kind, layers, ln in (("enc", self.enc, "ln2"), ("dec", self.dec, "ln3")):
                for i, lw in enumerate(layers):
                    if "w1_s" not in lw and "wqkv_s" not in lw:
                        continue
                    bounds = split16_operand_bounds(lw, ln, d)
                    keys = (["ffn_in", "ffn_hidden"] if "w1_s" in lw else []) + \
                           (["proj_in", "attn_context"] if "wqkv_s" in lw else [])
                    worst = max(keys, key=lambda k: bounds[k])
                    if bounds[worst] >= 65504.0:
                        raise ValueError(f"split16 is not usable for this model: {kind} layer {i}: |{worst}| can reach "
                                         f"{bounds[worst]:.3g}, beyond fp16's range (use ffn_dtype / proj_dtype float32)")
            if proj_dtype == "split16" and self.enc and "wqkv_s" in self.enc[0]:
                # ... and the tiled GEMMs of the encoder stage, which split BOTH operands on the fly (SC_GEMM_SPLIT16 in
                # csrc/streams.hip): conv2 reads relu(conv1), the subsampling Linear relu(conv2), the CTC / cross-K|V
                # projections the after_norm output.  A log-mel feature lies in [log 1e-10, log FLT_MAX] before the MVN.
                mean, std = self.mean64.abs().cpu().numpy(), self.std64.abs().cpu().numpy()
                feat = float(np.max((88.8 + mean) / np.maximum(std, 1e-30)))
                c1 = float(self.conv1_w.abs().sum(1).max()) * feat + float(self.conv1_b.abs().max())
                c2 = float(self.conv2_w.abs().sum(1).max()) * c1 + float(self.conv2_b.abs().max())
                enc_out = float((np.sqrt(d - 1.0) * self.enc_norm_g.abs() + self.enc_norm_b.abs()).max())
                for name, v in (("relu(conv1)", c1), ("relu(conv2)", c2), ("encoder output", enc_out)):
                    if v >= 65504.0:
                        raise ValueError(f"split16 projections are not usable for this model: |{name}| can reach {v:.3g}, "
                                         "beyond fp16's range (use proj_dtype float32)")
        self.dec_norm_g = dev(g("decoder.after_norm.weight"))
        self.dec_norm_b = dev(g("decoder.after_norm.bias"))
        self.out_w = dev(g("decoder.output_layer.weight"))
        self.out_b = dev(g("decoder.output_layer.bias"))
        # output layer in lane order (projected by the last layer's reduce kernel) when V is a multiple of d
        self.out_w_q = (pack_lane_weight(self.out_w) if d in PANEL_DIMS and cfg.vocab_size % d == 0
                        and ffn_fused_supported(d, cfg.ffn_dim) else None)
        self.out_w_qh = (self.out_w_q.to(torch.float16).contiguous()
                         if dec_dtype == "float16" and self.out_w_q is not None and d in (128, 256) else None)
        self.ctc_w = dev(g("ctc.ctc_lo.weight"))
        self.ctc_b = dev(g("ctc.ctc_lo.bias"))

    # ---- the flat, named view the stream-level C ABI takes (include/scasr.h: sc_engine_create / sc_engine_load)
    _TOP = ("window", "mel_fb", "twiddle", "pe", "mean64", "std64", "conv1_w", "conv1_b", "conv2_w", "conv2_b",
            "sub_out_w", "sub_out_b", "enc_norm_g", "enc_norm_b", "embed", "dec_norm_g", "dec_norm_b", "out_w", "out_b",
            "out_w_q", "out_w_qh", "ctc_w", "ctc_b")

    def named_tensors(self):
        """[(name, tensor)]: "window", ..., "enc.{i}.{field}", "dec.{i}.{field}" (fields = the dict keys above)."""
        out = [(n, getattr(self, n)) for n in self._TOP if getattr(self, n, None) is not None]
        for pre, layers in (("enc", self.enc), ("dec", self.dec)):
            for i, lw in enumerate(layers):
                out.extend((f"{pre}.{i}.{k}", v) for k, v in lw.items())
        return out

    def mvn_mode(self) -> int:
        return 0 if not self.has_mvn else (2 if self.mvn_is_f64 else 1)

    def save_packed(self, path):
        """Packed model file for hosts without Python / torch (sc_engine_load): 'SCPK1', sc_config, named tensors in
        the layouts above (weights already permuted / fragment-packed; fp32, MVN statistics fp64)."""
        import struct
        from ._abi import Config
        import ctypes as C
        cfg = self.cfg
        c = Config()
        for n, _ in Config._fields_:
            if n == "mvn_mode":
                c.mvn_mode = self.mvn_mode()
            else:
                setattr(c, n, getattr(cfg, n))
        ts = self.named_tensors()
        with open(path, "wb") as f:
            f.write(b"SCPK1\0\0\0")
            f.write(struct.pack("<i", C.sizeof(Config)))
            f.write(bytes(c))
            f.write(struct.pack("<i", len(ts)))
            for name, t in ts:
                nb = name.encode()
                a = t.detach().cpu().contiguous().numpy()
                f.write(struct.pack("<i", len(nb)))
                f.write(nb)
                code = {np.dtype(np.float64): 1, np.dtype(np.float16): 2}.get(a.dtype, 0)
                f.write(struct.pack("<iq", code, a.size))
                f.write(a.astype((np.float32, np.float64, np.float16)[code]).tobytes())
        return path

    def n_bytes(self, part="all") -> int:
        def tot(ds):
            return sum(t.numel() * t.element_size() for dct in ds for t in dct.values())
        enc = tot(self.enc) + sum(t.numel() * 4 for t in (self.conv1_w, self.conv1_b, self.conv2_w, self.conv2_b,
                                                            self.sub_out_w, self.sub_out_b, self.enc_norm_g, self.enc_norm_b))
        dec = tot(self.dec) + sum(t.numel() * 4 for t in (self.embed, self.dec_norm_g, self.dec_norm_b, self.out_w,
                                                           self.out_b, self.ctc_w, self.ctc_b))
        return {"enc": enc, "dec": dec, "all": enc + dec}[part]
