"""Model / search configuration of the streaming ASR hot path.

The fields mirror what the reference reads from ``config.yaml`` plus the
constants it bakes into code (SURVEY.md section 5 "Config / flags"):

* ``encoder_conf{output_size, attention_heads, num_blocks}``,
  ``decoder_conf{attention_heads, num_blocks}``,
  ``frontend_conf{n_fft, hop_length, win_length}``
  (reference: speechcatcher/speech2text_streaming.py:209-232)
* linear_units 2048, block/hop/look-ahead 40/16/16 are NOT read from the
  yaml by the reference (speechcatcher/model/espnet_asr_model.py:203,206 and
  speechcatcher/model/encoder/contextual_block_transformer_encoder.py:69-71).
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class ModelConfig:
    vocab_size: int = 1024
    n_mels: int = 80
    d_model: int = 256
    enc_heads: int = 8
    enc_layers: int = 30
    dec_heads: int = 8
    dec_layers: int = 14
    ffn_dim: int = 2048
    n_fft: int = 512
    hop_length: int = 160
    win_length: int = 400
    sample_rate: int = 16000
    block_size: int = 40
    hop_size: int = 16
    look_ahead: int = 16
    subsample: int = 4
    ln_eps: float = 1e-12
    pe_max_len: int = 5000

    @property
    def conv_freq1(self) -> int:
        return (self.n_mels - 3) // 2 + 1

    @property
    def conv_freq2(self) -> int:
        return (self.conv_freq1 - 3) // 2 + 1

    @property
    def blank_id(self) -> int:
        return 0

    @property
    def sos_id(self) -> int:
        # reference: speechcatcher/beam_search/beam_search.py:910-913
        return self.vocab_size - 1

    @property
    def eos_id(self) -> int:
        return self.vocab_size - 1

    def to_dict(self):
        return asdict(self)


# de_streaming_transformer_xl dims (docs/implementation/weight-loading.md:10-50,
# SURVEY.md section 0).
XL = ModelConfig()

# The reference's no-config defaults (speech2text_streaming.py:223-227,238-240: output_size 256, 4 heads, 12 + 6
# blocks) - what a model directory without encoder_conf / decoder_conf builds, and the best available stand-in for
# the `_m` checkpoints (BASELINE configs[0]; their config.yaml is not available offline).  Head dim 64.
M_DEFAULTS = ModelConfig(d_model=256, enc_heads=4, enc_layers=12, dec_heads=4, dec_layers=6)

# Stand-in for the `_l` checkpoints (BASELINE configs[3], en_streaming_transformer_l): their config.yaml is not available
# offline and only output_size / attention_heads / num_blocks are read from it (speech2text_streaming.py:215-232).
# ASSUMED: between the no-config defaults (256 / 4 heads / 12 + 6 blocks) and XL (256 / 8 heads / 30 + 14 blocks) -
# 256 / 4 heads (head dim 64) / 18 + 8 blocks.  The engine is generic in these three; nothing else varies.
L_LIKE = ModelConfig(d_model=256, enc_heads=4, enc_layers=18, dec_heads=4, dec_layers=8)

# Small model used for full-tensor golden fixtures.
TINY = ModelConfig(d_model=64, enc_heads=4, enc_layers=2, dec_heads=4,
                   dec_layers=2, ffn_dim=2048)

# Even smaller FFN for fast unit tests of the host logic.
MICRO = ModelConfig(d_model=32, enc_heads=2, enc_layers=2, dec_heads=2,
                    dec_layers=2, ffn_dim=2048)


@dataclass(frozen=True)
class SearchConfig:
    """Blockwise-synchronous beam search constants.

    reference: speechcatcher/beam_search/beam_search.py:276-341 (ctor
    defaults), :75 (pre-beam 40), speechcatcher/speechcatcher.py:221
    (ctc_weight 0.3).
    """
    beam_size: int = 10
    ctc_weight: float = 0.3
    pre_beam: int = 40
    max_length: int = 500
    use_bbd: bool = False

    @property
    def decoder_weight(self) -> float:
        return 1.0 - self.ctc_weight
