"""ctypes binding of libscasr.so (the C ABI declared in include/scasr.h).

cffi is not installed in the target image, so the
"thin C-ABI cffi layer" of the north star is ctypes in ABI mode; the binding
is plain C, nothing torch-specific crosses it.

The library is built in-tree (speechcatcher_amd/libscasr.so) by
``build()`` / ``__graft_entry__.build()``; loading fails LOUDLY when it is
missing - there is no CPU fallback in the product path.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libscasr.so"
CSRC = HERE / "csrc"

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
c_double_p = C.POINTER(C.c_double)
vp = C.c_void_p


class EncLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo",
                                  "ln2_g", "ln2_b", "w1", "b1", "w2", "b2", "w1_p", "w2_p",
                                  "wqkv_p", "wo_p", "w1_h", "w2_h", "wqkv_h", "wo_h", "w1_s", "w2_s",
                                  "wqkv_s", "wo_s")]


class DecLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo",
                                  "ln2_g", "ln2_b", "wq", "bq", "wo2", "bo2",
                                  "ln3_g", "ln3_b", "w1", "b1", "w2", "b2",
                                  "wo_p", "wq_p", "wo2_p", "w1_p", "w2_p", "wqkv_q", "wqkv_pp", "wq_pp", "wo_pp", "wo2_pp", "w1_h", "w2_h",
                                  "w1_s", "w2_s", "wqkv_pph", "wq_pph", "wo_pph", "wo2_pph")]


class Search(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("S", "W", "K", "V", "d", "H", "F", "n_layers", "TCAP", "LCAP",
                                  "blank", "eos", "sos")]
        + [(n, C.c_float) for n in ("w_dec", "w_ctc", "ln_eps")]
        + [(n, vp) for n in ("ctrl", "flags", "ctcx", "ckv", "skv", "yseq", "xpos", "anc", "score",
                             "sc_dec", "sc_ctc", "ctc_r", "ctc_s", "ctc_rnew", "dx", "dxn", "dqkv",
                             "datt", "dq", "dffh", "logits", "logp", "pre_ids", "psi", "psi_eos",
                             "cand_score", "cand_tok", "cand_ctc", "sel", "embed", "pe",
                             "dec_norm_g", "dec_norm_b", "out_w", "out_b", "layers", "rowmap")]
        + [("n_rows", C.c_int32), ("out_w_q", vp), ("ph1", vp), ("ph2", vp), ("ffn_part", vp),
           ("max_ffn_part", C.c_int32), ("tct", C.c_int32), ("ctcxT", vp), ("kv_half", C.c_int32), ("stat_rows", vp),
           ("out_w_qh", vp), ("act_half", C.c_int32), ("kv_rows", C.c_int32), ("kvflags", vp), ("ctc_rs", vp)]
    )


class Config(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("d_model", "enc_heads", "enc_layers", "dec_heads", "dec_layers", "ffn_dim",
                                           "vocab_size", "n_mels", "n_fft", "win_length", "hop_length", "sample_rate",
                                           "block_size", "hop_size", "look_ahead", "subsample", "conv_freq1",
                                           "conv_freq2", "blank_id", "sos_id", "eos_id", "pe_max_len", "mvn_mode")]
                + [("ln_eps", C.c_float)])


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", vp), ("numel", C.c_int64), ("dtype", C.c_int32)]


class StreamOptions(C.Structure):
    _fields_ = [("n_streams", C.c_int32), ("beam_size", C.c_int32), ("ctc_weight", C.c_float), ("use_bbd", C.c_int32),
                ("max_frames", C.c_int32), ("max_tokens", C.c_int32), ("pcm_capacity", C.c_int32),
                ("max_chunk_samples", C.c_int32), ("strict_reference", C.c_int32), ("kv_half", C.c_int32),
                ("kv_pool_rows", C.c_int32)]


class StreamInfo(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("enc_frames", "processed_block", "process_idx", "n_hyp", "hyp_len",
                                           "pcm_buffered", "frontend_started")] + [("decode_steps", C.c_int64)])


_SIGS = {
    "sc_last_error": (C.c_char_p, []),
    "sc_version": (C.c_int, []),
    "sc_gemm": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                          C.c_int, C.c_int, vp]),
    "sc_gemm_ln": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, vp, vp, C.c_float, vp, C.c_int, vp]),
    "sc_proj_ln_proj": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_float, vp, C.c_int,
                                  vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "sc_proj_ln_proj_supported": (C.c_int, [C.c_int]),
    "sc_pack_panel_weight": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "sc_pack_lane_weight": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "sc_ffn_ln": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp]),
    "sc_ffn_ln_h": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp]),
    "sc_ffn_ln_s": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp]),
    "sc_ffn_ln_proj_s": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp,
                                   vp, vp, vp, C.c_int, vp]),
    "sc_ffn_ln_proj_h": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp,
                                   vp, vp, vp, C.c_int, vp]),
    "sc_ffn_ln_supported": (C.c_int, [C.c_int, C.c_int]),
    "sc_rowtile_proj": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp, C.c_int, vp, vp,
                                  C.c_int, vp, vp, vp, vp]),
    "sc_rowtile_proj_h": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp, C.c_int, vp, vp,
                                    C.c_int, vp, vp, vp, vp]),
    "sc_rowtile_proj_s": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp, C.c_int, vp, vp,
                                    C.c_int, vp, vp, vp, vp]),
    "sc_rowtile_proj_supported": (C.c_int, [C.c_int, C.c_int]),
    "sc_workspace_bytes": (C.c_size_t, [vp]),
    "sc_ffn_ln_proj": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp,
                                 vp, vp, vp, C.c_int, vp]),
    "sc_graph_capture_begin": (C.c_int, [vp]),
    "sc_graph_capture_end": (C.c_int, [vp, C.POINTER(vp)]),
    "sc_graph_launch": (C.c_int, [vp, vp]),
    "sc_graph_destroy": (C.c_int, [vp]),
    "sc_set_workspace": (C.c_int, [vp, C.c_size_t]),
    "sc_set_stream_workspace": (C.c_int, [vp, vp, C.c_size_t]),
    "sc_prof_collect_kinds": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_longlong), C.c_int]),
    "sc_prof_enable": (C.c_int, [C.c_int]),
    "sc_prof_collect": (C.c_int, [c_double_p, c_double_p, C.POINTER(C.c_longlong)]),
    "sc_prof_collect2": (C.c_int, [c_double_p, c_double_p, c_double_p, C.POINTER(C.c_longlong)]),
    "sc_prof_event_overhead_ms": (C.c_double, [vp]),
    "sc_layernorm": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp]),
    "sc_copy_rows": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "sc_log_softmax_rows": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "sc_logmel": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int,
                            C.c_int, C.c_int, C.c_int, vp, vp]),
    "sc_conv1": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]),
    "sc_block_pack": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]),
    "sc_ctx_handoff": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "sc_enc_attention": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "sc_encoder_layers": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp,
                                    vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "sc_glu_dwconv_bn_swish": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp,
                                         C.c_float, vp, vp]),
    "sc_relpos_attention": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "sc_relpos_attention_masked": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "sc_ctc_extend_state": (C.c_int, [vp, vp]),
    "sc_dec_embed": (C.c_int, [vp, vp]),
    "sc_kv_rows_to_half": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "sc_dec_self_attn": (C.c_int, [vp, C.c_int, vp]),
    "sc_dec_cross_attn": (C.c_int, [vp, C.c_int, vp]),
    "sc_decoder_layers": (C.c_int, [vp, vp]),
    "sc_logsoftmax_topk": (C.c_int, [vp, vp]),
    "sc_ctc_prefix_scan": (C.c_int, [vp, vp]),
    "sc_fuse_topw": (C.c_int, [vp, vp]),
    "sc_beam_prune": (C.c_int, [vp, vp]),
    "sc_ctc_gather_state": (C.c_int, [vp, vp]),
    "sc_ctc_gather_state_split": (C.c_int, [vp, C.c_int, vp]),
    "sc_decode_step": (C.c_int, [vp, vp]),
    "sc_decode_step_ex": (C.c_int, [vp, C.c_int, vp]),
    "sc_ctc_prefix_scan_split": (C.c_int, [vp, C.c_int, vp]),
    "sc_dec_layer_fused_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sc_dec_layer_self": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, vp]),
    "sc_dec_layer_cross": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "sc_dec_layer_ffn": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int), vp]),
    "sc_dec_output_logits": (C.c_int, [vp, vp, vp, vp, C.c_int, vp]),
    "sc_streams_kv_rows": (C.c_int, [vp]),
    "sc_dec_layer_stream_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sc_dec_layer_stream": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "sc_dec_layer_reduce_ln": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "sc_dec_layer_ffn_xn": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int), vp]),
    # stream-level API
    "sc_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(NamedTensor), C.c_int, C.c_int, C.POINTER(vp)]),
    "sc_engine_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "sc_engine_config": (C.c_int, [vp, C.POINTER(Config)]),
    "sc_engine_destroy": (None, [vp]),
    "sc_streams_create": (C.c_int, [vp, C.POINTER(StreamOptions), C.POINTER(vp)]),
    "sc_streams_destroy": (None, [vp]),
    "sc_push": (C.c_int, [vp, c_int_p, C.POINTER(vp), c_int_p, C.POINTER(C.c_uint8), C.c_int, c_int_p]),
    "sc_push_features": (C.c_int, [vp, c_int_p, C.POINTER(vp), c_int_p, C.POINTER(C.c_uint8), C.c_int, c_int_p]),
    "sc_get_hyps": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_double_p, c_double_p,
                              c_double_p]),
    "sc_get_hyps_batch": (C.c_int, [vp, c_int_p, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p,
                                    c_double_p, c_double_p, c_double_p]),
    "sc_submit": (C.c_int, [vp, c_int_p, C.POINTER(vp), c_int_p, C.POINTER(C.c_uint8), C.c_int]),
    "sc_poll": (C.c_int, [vp, C.c_int, C.c_int, c_int_p, c_int_p]),
    "sc_streams_outstanding": (C.c_int, [vp]),
    "sc_streams_set_encoder_batch": (C.c_int, [vp, C.c_int]),
    "sc_streams_set_queue_depth": (C.c_int, [vp, C.c_int]),
    "sc_stream_last_error": (C.c_char_p, [vp, C.c_int]),
    "sc_reset": (C.c_int, [vp, C.c_int]),
    "sc_stream_info": (C.c_int, [vp, C.c_int, C.POINTER(StreamInfo)]),
    "sc_streams_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "sc_streams_capture_stats": (C.c_int, [vp, C.POINTER(C.c_long), c_double_p]),
    "sc_streams_set_graphs": (C.c_int, [vp, C.c_int]),
    "sc_marker": (C.c_int, [C.c_int, vp]),
    "sc_streams_host_times": (C.c_int, [vp, c_double_p, c_double_p]),
    "sc_streams_bucket_times": (C.c_int, [vp, c_double_p, C.POINTER(C.c_long)]),
    "sc_streams_take_xattn_rows": (C.c_long, [vp]),
    "sc_streams_take_xattn_rows_by_kernel": (C.c_int, [vp, vp]),
    "sc_streams_take_attn_counters": (C.c_int, [vp, vp]),
    "sc_streams_hip_stream": (vp, [vp]),
    "sc_streams_pcm": (vp, [vp, C.POINTER(C.c_long)]),
    "sc_streams_write_pcm": (C.c_int, [vp, C.c_int, C.c_long, vp, C.c_long]),
    "sc_streams_read_pcm_buffer": (C.c_long, [vp, C.c_int, vp, C.c_long]),
    "sc_streams_read_enc": (C.c_int, [vp, C.c_int, vp, C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())

# revision of include/scasr.h these ctypes mirrors were written against (SC_ABI_VERSION): a library built from another
# revision would be handed mis-laid-out structs
ABI_VERSION = 6

_lib = None


class ScasrError(RuntimeError):
    pass


def build(verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into speechcatcher_amd/libscasr.so."""
    cmd = ["make", "-C", str(CSRC), "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise ScasrError("building libscasr.so failed")
    return LIB_PATH


def load():
    """Load the library; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; import it FIRST so that libscasr
    # binds to the same HIP runtime instance (two runtimes in one process do
    # not see each other's device / streams).
    import torch  # noqa: F401
    lib_path = LIB_PATH
    # A/B runs of library VARIANTS (tools/ab_libs.sh, tools/build_variant.sh): honoured only with the test hooks on - a
    # production process always loads the in-tree product build (the variants used to be copied over it)
    if os.environ.get("SC_TEST_HOOKS") == "1" and os.environ.get("SC_LIB_VARIANT"):
        lib_path = Path(os.environ["SC_LIB_VARIANT"]).resolve()
    if not lib_path.exists():
        raise ScasrError(
            f"{lib_path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(str(lib_path))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.sc_version()
    if got != ABI_VERSION:
        raise ScasrError(f"{lib_path} was built from ABI revision {got} of include/scasr.h, this binding is revision "
                         f"{ABI_VERSION}: rebuild the library (`python -c 'import __graft_entry__ as g; g.build()'`)")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().sc_last_error()
        raise ScasrError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
