"""Drop-in ``Speech2TextStreaming`` / ``load_model`` for speechcatcher's native
decoder path, backed by the MI355X HIP engine.

Mirrors the reference's public surface:

* ``Speech2TextStreaming(model_dir, beam_size=5, ctc_weight=0.3, device, dtype,
  use_bbd=False)``                      speechcatcher/speech2text_streaming.py:43-51
* ``__call__(speech, is_final=False, finalize_all=False)`` -> list of
  ``(text, tokens, token_ids)``           :402-539
* ``reset / recognize / recognize_stream / n_best_hypotheses /
  get_best_hypothesis``                   :252-263, :541-596
* ``load_model(tag, device, beam_size, quiet, cache_dir, decoder_impl, fp16,
  use_bbd)``                              speechcatcher/speechcatcher.py:126-227

There is no CPU path: constructing the object without a ROCm GPU or without
the built HIP library raises.
"""
import logging
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _abi
from .config import ModelConfig, SearchConfig
from .engine import EngineError
from .weights import PackedWeights

logger = logging.getLogger(__name__)

_CKPT_NAMES = ("valid.acc.best.pth", "valid.acc.ave_6best.pth", "valid.acc.ave.pth", "model.pth", "checkpoint.pth")


def find_checkpoint(model_dir: Path) -> Path:
    """speech2text_streaming.py:163-189"""
    paths = [model_dir / n for n in _CKPT_NAMES]
    for exp_dir in model_dir.glob("exp/*/"):
        paths.extend(exp_dir / n for n in _CKPT_NAMES)
    for p in paths:
        if p.exists():
            return p
    raise FileNotFoundError(f"No checkpoint found in {model_dir}")


def load_state_dict(ckpt: Path):
    ck = torch.load(ckpt, map_location="cpu")
    if isinstance(ck, dict) and "model" in ck:
        return ck["model"]
    if isinstance(ck, dict) and "state_dict" in ck:
        return ck["state_dict"]
    return ck


def config_from_dir(model_dir: Path, sd) -> ModelConfig:
    """speech2text_streaming.py:195-242: vocab from the embedding, arch from
    config.yaml (only output_size / attention_heads / num_blocks / frontend
    sizes are read; FFN 2048 and block 40/16/16 are fixed)."""
    import yaml

    if "decoder.embed.0.weight" in sd:
        vocab = sd["decoder.embed.0.weight"].shape[0]
    elif "decoder.output_layer.weight" in sd:
        vocab = sd["decoder.output_layer.weight"].shape[0]
    else:
        raise ValueError("Could not infer vocab_size from checkpoint")
    enc = dec = fe = {}
    cp = model_dir / "config.yaml"
    if cp.exists():
        with open(cp) as f:
            conf = yaml.safe_load(f) or {}
        enc = conf.get("encoder_conf", {}) or {}
        dec = conf.get("decoder_conf", {}) or {}
        fe = conf.get("frontend_conf", {}) or {}
        return ModelConfig(
            vocab_size=vocab, d_model=enc.get("output_size", 256), enc_heads=enc.get("attention_heads", 4),
            enc_layers=enc.get("num_blocks", 12), dec_heads=dec.get("attention_heads", 4),
            dec_layers=dec.get("num_blocks", 6), n_fft=fe.get("n_fft", 512),
            hop_length=fe.get("hop_length", 160), win_length=fe.get("win_length", 400))
    return ModelConfig(vocab_size=vocab, d_model=256, enc_heads=4, enc_layers=12, dec_heads=4, dec_layers=6)


def load_stats(model_dir: Path):
    """speech2text_streaming.py:76-95 + checkpoint_loader.py:210-237"""
    cands = [
        model_dir / "feats_stats.npz",
        model_dir.parent / "asr_stats_raw_de_bpe1024/train/feats_stats.npz",
        model_dir.parent.parent / "asr_stats_raw_de_bpe1024/train/feats_stats.npz",
        model_dir / "../stats/train/feats_stats.npz",
    ]
    for p in cands:
        if p.exists():
            st = np.load(p)
            if "mean" in st:
                return st["mean"], st["std"]
            count = st["count"]
            mean = st["sum"] / count
            std = np.sqrt(np.maximum(st["sum_square"] / count - mean ** 2, 1e-10))
            return mean, std
    logger.warning("Normalization stats not found")
    return None, None


def load_token_list(model_dir: Path) -> Optional[List[str]]:
    """speech2text_streaming.py:100-124: ["<blank>", SP[0]] + SP[3:] + ["<sos/eos>"]"""
    try:
        import sentencepiece as spm
    except ImportError:
        return None
    for p in (model_dir / "bpe.model",
              model_dir.parent.parent / "data/de_token_list/bpe_unigram1024/bpe.model",
              model_dir / "../data/de_token_list/bpe_unigram1024/bpe.model"):
        if p.exists():
            sp = spm.SentencePieceProcessor()
            sp.Load(str(p))
            n = sp.GetPieceSize()
            return ["<blank>", sp.IdToPiece(0)] + [sp.IdToPiece(i) for i in range(3, n)] + ["<sos/eos>"]
    return None


# ---------------------------------------------------------------------------
# single-file model blob (offline-friendly asset path)
# ---------------------------------------------------------------------------
BLOB_SUFFIX = ".scasr"
_CFG_FIELDS = ("vocab_size", "d_model", "enc_heads", "enc_layers", "dec_heads", "dec_layers", "n_fft",
               "hop_length", "win_length")


def save_model_blob(model_dir: Union[str, Path], out_path: Union[str, Path]) -> Path:
    """Pack an ESPnet-layout model directory (``*.pth`` + ``config.yaml`` +
    ``feats_stats.npz`` + ``bpe.model``: speech2text_streaming.py:76-81,100-105,
    163-180) into ONE file that needs neither yaml, numpy archives nor
    sentencepiece at load time: fp32 state dict, architecture, MVN statistics in
    their original precision (float64 for the sum/count form, A11) and the token
    list.  Loadable with ``torch.load(weights_only=True)``."""
    model_dir, out_path = Path(model_dir), Path(out_path)
    sd = load_state_dict(find_checkpoint(model_dir))
    cfg = config_from_dir(model_dir, sd)
    mean, std = load_stats(model_dir)
    blob = {"format": 1, "config": {k: int(getattr(cfg, k)) for k in _CFG_FIELDS},
            "state_dict": {k: v.detach().to(torch.float32).contiguous() for k, v in sd.items()
                           if isinstance(v, torch.Tensor) and v.dtype.is_floating_point},
            "mean": None if mean is None else torch.from_numpy(np.ascontiguousarray(mean)),
            "std": None if std is None else torch.from_numpy(np.ascontiguousarray(std)),
            "token_list": load_token_list(model_dir)}
    torch.save(blob, out_path)
    return out_path


def load_model_blob(path: Union[str, Path]):
    """-> (state_dict, ModelConfig, mean, std, token_list)"""
    blob = torch.load(Path(path), map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or blob.get("format") != 1:
        raise ValueError(f"{path} is not a speechcatcher_amd model blob")
    cfg = ModelConfig(**blob["config"])
    mean = None if blob["mean"] is None else blob["mean"].numpy()
    std = None if blob["std"] is None else blob["std"].numpy()
    return blob["state_dict"], cfg, mean, std, blob["token_list"]


class _BeamSearchView:
    """The attributes callers/tests touch on ``s2t.beam_search``
    (tests/test_speech2text_streaming.py:213-219)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def processed_block(self):
        return self._o.batch.st[self._o.stream].processed_block

    @property
    def process_idx(self):
        return self._o.batch.st[self._o.stream].process_idx

    @property
    def encoder_buffer(self):
        enc = self._o.batch.encoder_buffer(self._o.stream)
        return None if enc is None else torch.from_numpy(enc).unsqueeze(0)

    def reset(self):
        self._o.batch.reset(self._o.stream)


class _BeamStateView:
    def __init__(self, hyps):
        self.hypotheses = hyps
        self.output_index = 0


def hyps_to_results(hyps, is_final, finalize_all, token_list, fmt="native"):
    """speech2text_streaming.py:466-539 (EOS id hard-coded 1023: A4;
    non-final calls use a fresh output_index 0: A5)."""
    if not is_final or not finalize_all:
        out = [h for h in hyps if h["yseq"][-1] == 1023]
    else:
        out = hyps
    res = []
    for h in out:
        ys, xp = h["yseq"], h["xpos"]
        if is_final:
            ids, pos = ys[1:], xp[1:]
            if ids and ids[-1] == 1023:
                ids, pos = ids[:-1], pos[:-1]
        else:
            end = min(0 + 1, len(ys))
            ids, pos = ys[1:end], xp[1:end]
            if ids and ids[-1] == 1023:
                ids, pos = ids[:-1], pos[:-1]
        keep = [(t, p) for t, p in zip(ids, pos) if t not in (0, 1, 1023)]
        ids = [t for t, _ in keep]
        pos = [p for _, p in keep]
        if token_list is not None:
            toks = [token_list[t] for t in ids]
            text = "".join(toks).replace("▁", " ").strip()
        else:
            toks = [str(t) for t in ids]
            text = " ".join(toks)
        if fmt == "espnet":
            # (text, token, token_int, token_pos, hyp): asr_inference_streaming.py:364
            res.append((text, toks, ids, pos, h))
        else:
            # A17: the reference builds token_ids with a list comprehension over a LongTensor, so the
            # native 3-tuples carry 0-dim torch.LongTensors, not ints (speech2text_streaming.py:498,518,537)
            res.append((text, toks, [torch.tensor(t, dtype=torch.long) for t in ids]))
    return res


class Speech2TextStreaming:
    def __init__(self, model_dir: Union[str, Path], beam_size: int = 5, ctc_weight: float = 0.3,
                 device: str = "cuda", dtype: str = "float32", use_bbd: bool = False,
                 max_frames: int = 4800, max_tokens: int = 1024, result_format: str = "native",
                 max_chunk_samples: int = 480000, strict_reference: bool = True, _shared=None):
        """``max_chunk_samples``: longest single call (the reference accepts any length per call and its frame
        arithmetic depends on the call boundaries, so a long call cannot be split internally; the scratch for one
        call is sized from this: 30 s by default, ~80 MB).  ``strict_reference``: reset() behaves like the
        reference's (stale CTC table, A13 counter: StreamBatch.reset); False gives a clean stream.
        ``dtype="float16"`` (the reference keeps fp32 weights under autocast for this constructor argument,
        speech2text_streaming.py:57,64-70): fp16 feed-forward weights with fp16 MFMA inputs and fp16 K|V caches, every
        sum / softmax / LayerNorm / score in fp32 (BASELINE configs[4]; `load_model(fp16=True)` stays fp32 with the
        reference's warning).  ``dtype="split16"`` (no counterpart in the reference): fp32 results - every hypothesis of
        the reference fixtures at the fp32 tolerance - with the feed-forward, encoder-projection and tiled-GEMM product
        sums evaluated on the fp16 matrix pipe from fp16 hi + lo splits of both operands."""
        self.model_dir = Path(model_dir)
        self.beam_size = beam_size
        self.ctc_weight = ctc_weight
        self.device = "cuda:0" if device in ("cuda", "gpu") else device
        if not str(self.device).startswith("cuda"):
            raise _abi.ScasrError(
                "speechcatcher_amd has no CPU path; pass device='cuda' (use the reference package for CPU)")
        if dtype not in ("float32", "float16", "split16"):
            logger.warning("dtype %s requested; using float32", dtype)
            dtype = "float32"
        self.dtype = torch.float32     # activations, scores and results
        half = dtype == "float16"
        self.use_bbd = use_bbd
        self.result_format = result_format
        if self.model_dir.is_file() and self.model_dir.suffix == BLOB_SUFFIX:
            sd, self.cfg, self.mean, self.std, self.token_list = load_model_blob(self.model_dir)
        else:
            sd = load_state_dict(find_checkpoint(self.model_dir))
            self.cfg = config_from_dir(self.model_dir, sd)
            self.mean, self.std = load_stats(self.model_dir)
            self.token_list = load_token_list(self.model_dir)
        self.weights = PackedWeights(sd, self.cfg, self.device, self.mean, self.std,
                                     ffn_dtype=dtype, proj_dtype=dtype,
                                     dec_dtype="float16" if half else "float32")
        self.model = self.weights
        # the decoder itself: the C++ engine behind the stream-level C ABI (csrc/streams.hip); raises without
        # the built library or without a GPU - there is no fallback
        from .native import NativeStreamBatch
        try:
            self.batch = NativeStreamBatch(self.weights, 1,
                                           SearchConfig(beam_size=beam_size, ctc_weight=ctc_weight, use_bbd=use_bbd),
                                           max_frames=max_frames, max_tokens=max_tokens,
                                           pcm_capacity=max(1 << 20, 2 * max_chunk_samples),
                                           max_chunk_samples=max_chunk_samples, strict_reference=strict_reference,
                                           kv_dtype="float16" if half else "float32")
        except EngineError as e:
            raise _abi.ScasrError(str(e)) from e
        self.stream = 0
        self.win_length = self.cfg.win_length
        self.hop_length = self.cfg.hop_length
        self.beam_search = _BeamSearchView(self)
        self.reset()

    # ------------------------------------------------------------------
    def reset(self):
        self.batch.reset(self.stream)
        self.beam_state = None
        self.processed_frames = 0

    @property
    def frontend_states(self):
        if not self.batch.st[self.stream].fe_started:
            return None
        return {"waveform_buffer": torch.from_numpy(self.batch.waveform_buffer(self.stream))}

    def __call__(self, speech, is_final: bool = False, finalize_all: bool = False,
                 always_assemble_hyps: bool = False):
        if isinstance(speech, torch.Tensor):
            speech = speech.detach().cpu().numpy()
        speech = np.asarray(speech, dtype=np.float32)
        if speech.ndim == 1:
            out = self.batch.push([(self.stream, speech, is_final)])
            if not out[self.stream]:
                return []
        elif speech.ndim == 2:
            x = speech
            if self.mean is not None and self.std is not None:
                x = ((x - self.mean) / self.std).astype(np.float32)
            self.batch.push_features([(self.stream, torch.from_numpy(np.ascontiguousarray(x)), is_final)])
        else:
            self.batch.push_features([(self.stream, torch.from_numpy(np.ascontiguousarray(speech[0])), is_final)])
        hyps = self.batch.hypotheses(self.stream)
        self.beam_state = _BeamStateView(hyps)
        return hyps_to_results(hyps, is_final, finalize_all, self.token_list, self.result_format)

    def recognize(self, speech):
        self.reset()
        return self(speech, is_final=True)

    def recognize_stream(self, chunks: Sequence):
        self.reset()
        results = None
        for i, chunk in enumerate(chunks):
            results = self(chunk, is_final=(i == len(chunks) - 1))
        return results if results is not None else []

    @property
    def n_best_hypotheses(self) -> int:
        return self.beam_size

    def get_best_hypothesis(self):
        if self.beam_state is None or not self.beam_state.hypotheses:
            return None
        res = hyps_to_results(self.beam_state.hypotheses, True, True, self.token_list, self.result_format)
        return res[0] if res else None


def create_streaming_interface(model_dir, beam_size: int = 5, ctc_weight: float = 0.3,
                               device: str = "cuda") -> Speech2TextStreaming:
    return Speech2TextStreaming(model_dir=model_dir, beam_size=beam_size, ctc_weight=ctc_weight, device=device)


# model tags of the reference (speechcatcher/speechcatcher.py:50-57)
TAGS = {
    "de_streaming_transformer_m": "speechcatcher/speechcatcher_german_espnet_streaming_transformer_13k_train_size_m_raw_de_bpe1024",
    "de_streaming_transformer_l": "speechcatcher/speechcatcher_german_espnet_streaming_transformer_13k_train_size_l_raw_de_bpe1024",
    "de_streaming_transformer_xl": "speechcatcher/speechcatcher_german_espnet_streaming_transformer_13k_train_size_xl_raw_de_bpe1024",
    "en_streaming_transformer_m": "speechcatcher/speechcatcher_english_espnet_streaming_transformer_m_raw_en_bpe1024",
    "en_streaming_transformer_l": "speechcatcher/speechcatcher_english_espnet_streaming_transformer_l_raw_en_bpe1024",
}


def load_model(tag, device="cuda", beam_size=5, quiet=False, cache_dir="~/.cache/espnet",
               decoder_impl="native", fp16=False, use_bbd=False):
    """speechcatcher.load_model (speechcatcher/speechcatcher.py:126-227).
    ``tag`` may be a model tag (needs espnet_model_zoo for the download, like
    the reference) or a local model directory."""
    if decoder_impl != "native":
        raise ValueError("speechcatcher_amd only implements --decoder native")
    if fp16:
        logger.warning("FP16 is not supported with the native decoder yet. Disabling FP16.")  # :205-210
    p = Path(str(tag)).expanduser()
    if p.is_dir() or (p.is_file() and p.suffix == BLOB_SUFFIX):
        model_dir = p
    else:
        try:
            from espnet_model_zoo.downloader import ModelDownloader
        except ImportError as e:
            raise ImportError("downloading model tags needs espnet_model_zoo (as in the reference); "
                              "pass a local model directory instead") from e
        info = ModelDownloader(str(Path(cache_dir).expanduser())).download_and_unpack(TAGS.get(tag, tag))
        model_dir = Path(info.get("asr_model_file", info.get("asr_train_config"))).parent
    return Speech2TextStreaming(model_dir, beam_size=beam_size, ctc_weight=0.3, device=device,
                                dtype="float32", use_bbd=use_bbd)
