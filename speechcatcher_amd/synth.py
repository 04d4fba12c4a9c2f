"""Seeded synthetic checkpoints in the ESPnet state-dict schema.

Real checkpoints are not available offline, so every
golden fixture, parity test and bench run uses weights from this generator.
The same generator runs in the survey container (to drive the imported
reference) and on the GPU box (to feed the HIP engine), so weights never have
to be shipped.

Schema: SURVEY.md Appendix B (reference:
speechcatcher/model/checkpoint_loader.py:134-149, names identical to ESPnet).
"""
from collections import OrderedDict
from pathlib import Path
from typing import Dict

import numpy as np
import torch

from .config import ModelConfig


def state_dict_schema(cfg: ModelConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    d, f, v = cfg.d_model, cfg.ffn_dim, cfg.vocab_size
    sch: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def lin(prefix, out_f, in_f):
        sch[prefix + ".weight"] = (out_f, in_f)
        sch[prefix + ".bias"] = (out_f,)

    def ln(prefix):
        sch[prefix + ".weight"] = (d,)
        sch[prefix + ".bias"] = (d,)

    def mha(prefix):
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            lin(f"{prefix}.{n}", d, d)

    sch["encoder.embed.conv.0.weight"] = (d, 1, 3, 3)
    sch["encoder.embed.conv.0.bias"] = (d,)
    sch["encoder.embed.conv.2.weight"] = (d, d, 3, 3)
    sch["encoder.embed.conv.2.bias"] = (d,)
    lin("encoder.embed.out", d, d * cfg.conv_freq2)
    for i in range(cfg.enc_layers):
        p = f"encoder.encoders.{i}"
        mha(p + ".self_attn")
        lin(p + ".feed_forward.w_1", f, d)
        lin(p + ".feed_forward.w_2", d, f)
        ln(p + ".norm1")
        ln(p + ".norm2")
    ln("encoder.after_norm")
    sch["decoder.embed.0.weight"] = (v, d)
    for i in range(cfg.dec_layers):
        p = f"decoder.decoders.{i}"
        mha(p + ".self_attn")
        mha(p + ".src_attn")
        lin(p + ".feed_forward.w_1", f, d)
        lin(p + ".feed_forward.w_2", d, f)
        ln(p + ".norm1")
        ln(p + ".norm2")
        ln(p + ".norm3")
    ln("decoder.after_norm")
    lin("decoder.output_layer", v, d)
    lin("ctc.ctc_lo", v, d)
    return sch


def make_state_dict(cfg: ModelConfig, seed: int = 1234) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 weights; one generator stream, schema order."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in state_dict_schema(cfg).items():
        is_norm = ".norm" in name or "after_norm" in name
        if is_norm and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif is_norm:
            t = 0.1 * torch.randn(shape, generator=g)
        elif name == "decoder.embed.0.weight":
            t = torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
        sd[name] = t.to(torch.float32).contiguous()
    return sd


def make_stats(cfg: ModelConfig, seed: int = 99, kind: str = "meanstd"):
    """Global-MVN statistics.  ``kind='sums'`` exercises the float64
    sum/sum_square/count path (reference:
    speechcatcher/model/checkpoint_loader.py:225-231)."""
    rng = np.random.RandomState(seed)
    mean = (-8.0 + rng.randn(cfg.n_mels)).astype(np.float64)
    std = (2.0 + 0.5 * rng.rand(cfg.n_mels)).astype(np.float64)
    if kind == "meanstd":
        return {"mean": mean.astype(np.float32), "std": std.astype(np.float32)}
    if kind == "unit":
        return {"mean": np.zeros(cfg.n_mels, np.float32), "std": np.ones(cfg.n_mels, np.float32)}
    count = np.float64(12345.0)
    return {"count": count, "sum": mean * count,
            "sum_square": (std ** 2 + mean ** 2) * count}


def stats_to_mean_std(stats: Dict[str, np.ndarray]):
    """reference: speechcatcher/model/checkpoint_loader.py:210-237"""
    if "mean" in stats:
        return stats["mean"], stats["std"]
    count = stats["count"]
    mean = stats["sum"] / count
    mean_square = stats["sum_square"] / count
    std = np.sqrt(np.maximum(mean_square - mean ** 2, 1e-10))
    return mean, std


def write_model_dir(path, cfg: ModelConfig, seed: int = 1234,
                    stats_kind: str = "meanstd") -> Path:
    """Lay out a model directory the way ``Speech2TextStreaming`` expects it
    (reference: speechcatcher/speech2text_streaming.py:76-81,163-180;
    fixture pattern of tests/test_speech2text_streaming.py:19-62)."""
    import yaml

    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    torch.save({"model": make_state_dict(cfg, seed)}, path / "model.pth")
    conf = {
        "encoder_conf": {"output_size": cfg.d_model, "attention_heads": cfg.enc_heads,
                         "num_blocks": cfg.enc_layers},
        "decoder_conf": {"attention_heads": cfg.dec_heads, "num_blocks": cfg.dec_layers},
        "frontend_conf": {"n_fft": cfg.n_fft, "hop_length": cfg.hop_length,
                          "win_length": cfg.win_length},
    }
    with open(path / "config.yaml", "w") as f:
        yaml.safe_dump(conf, f)
    np.savez(path / "feats_stats.npz", **make_stats(cfg, kind=stats_kind))
    return path


def synth_audio(stream_id: int, n_samples: int) -> np.ndarray:
    """Per-stream synthetic audio: N(0, 0.1^2) clipped to +-1, seed 1000+id
    (SURVEY.md section 8(d) "Synthetic inputs")."""
    g = torch.Generator().manual_seed(1000 + stream_id)
    x = torch.randn(n_samples, generator=g) * 0.1
    return x.clamp_(-1.0, 1.0).numpy().astype(np.float32)


def make_conformer_state(channels: int, n_head: int, kernel_size: int = 31, seed: int = 4321):
    """Seeded weights for the stand-alone Conformer blocks, in the reference's
    parameter names (ConvolutionModule: model/layers/convolution.py:46-80;
    RelPositionMultiHeadedAttention: model/attention/multi_head_attention.py:281-298)."""
    g = torch.Generator().manual_seed(seed)
    C, k, dk = channels, kernel_size, channels // n_head

    def u(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return ((torch.rand(shape, generator=g) * 2 - 1) * b).float()

    conv = OrderedDict()
    conv["layernorm.weight"] = 1.0 + 0.1 * torch.randn(C, generator=g)
    conv["layernorm.bias"] = 0.1 * torch.randn(C, generator=g)
    conv["pointwise_conv1.weight"] = u((2 * C, C, 1), C)
    conv["pointwise_conv1.bias"] = 0.05 * torch.randn(2 * C, generator=g)
    conv["depthwise_conv.weight"] = u((C, 1, k), k)
    conv["depthwise_conv.bias"] = 0.05 * torch.randn(C, generator=g)
    conv["batch_norm.weight"] = 1.0 + 0.1 * torch.randn(C, generator=g)
    conv["batch_norm.bias"] = 0.1 * torch.randn(C, generator=g)
    conv["batch_norm.running_mean"] = 0.1 * torch.randn(C, generator=g)
    conv["batch_norm.running_var"] = 0.5 + torch.rand(C, generator=g)
    conv["pointwise_conv2.weight"] = u((C, C, 1), C)
    conv["pointwise_conv2.bias"] = 0.05 * torch.randn(C, generator=g)
    att = OrderedDict()
    for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
        att[n + ".weight"] = u((C, C), C)
        att[n + ".bias"] = 0.05 * torch.randn(C, generator=g)
    att["linear_pos.weight"] = u((C, C), C)
    att["pos_bias_u"] = 0.2 * torch.randn(n_head, dk, generator=g)
    att["pos_bias_v"] = 0.2 * torch.randn(n_head, dk, generator=g)
    return ({k_: v.float().contiguous() for k_, v in conv.items()},
            {k_: v.float().contiguous() for k_, v in att.items()})


def synth_energy_curve(seed: int, n_frames: int) -> np.ndarray:
    """Seeded smoothed negative log-energy curve (100 frames/s) with speech-like
    bursts and pauses: input of the segment cut search fixtures
    (tools/gen_golden_segmenter.py, tests/test_segmenter.py)."""
    from scipy.ndimage import gaussian_filter1d
    rng = np.random.RandomState(seed)
    raw = rng.randn(n_frames) * 2.0 + 20.0 + 6.0 * np.sin(np.arange(n_frames) / rng.uniform(150, 400))
    for _ in range(n_frames // 900):
        a = rng.randint(0, n_frames - 60)
        raw[a:a + rng.randint(20, 120)] -= rng.uniform(4, 12)
    return gaussian_filter1d(raw, sigma=20) * -1.0
