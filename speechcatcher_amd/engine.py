"""Host side of the MI355X streaming engine: per-stream integer bookkeeping
(the three nested carry-over buffers, the block schedule, the blockwise-
synchronous beam-search control flow) for S concurrent streams, driving
batched device ops through a ``Backend`` (speechcatcher_amd.hip_backend in
the product; a torch spec backend in tests).

All arithmetic happens in the backend's kernels; this file only decides WHAT
to launch.  It replaces, for S streams at once:

* Speech2TextStreaming.apply_frontend        speechcatcher/speech2text_streaming.py:278-400
* ContextualBlockTransformerEncoder.forward_infer
                                             speechcatcher/model/encoder/contextual_block_transformer_encoder.py:241-419
* BlockwiseSynchronousBeamSearch.process_block / _decode_one_block
                                             speechcatcher/beam_search/beam_search.py:507-653,655-838
(paths relative to /root/reference).  SURVEY.md Appendix D/E describe the
exact buffering arithmetic restated here.

Role: the EXECUTABLE SPECIFICATION of the host logic.  The product engine is the C++
state machine behind the stream-level C ABI (csrc/streams.hip, speechcatcher_amd/native.py);
this class runs the same schedule in lock-step, run to completion, on the CPU spec backend
(oracle/kernel_spec.py) against the reference fixtures - which is how the host logic is
checked without a GPU - and on the HIP kernels for the per-kernel lock-step parity tests.
Continuous batching (sc_submit / sc_poll) exists in the C++ engine only.
"""
import copy
import math
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import SearchConfig
from .weights import PackedWeights

# flags returned by the beam_prune kernel
F_ANY_EOS, F_BEST_EOS, F_ALL_EOS, F_REPEAT = 1, 2, 4, 8
# ctrl columns (int32 [S, 8])
C_ACTIVE, C_CUR, C_FINAL, C_T, C_L, C_NHYP, C_HAS, C_TOLD = range(8)
C_TCTC = 7   # decode-step rows: length of the CTC table / states (include/scasr.h)


class EngineError(RuntimeError):
    pass


class StreamFault(Exception):
    """Internal carrier: planning of ONE stream failed with ``exc`` (an EngineError for a
    capacity limit, or the RuntimeError the reference itself dies with - A3).  push()
    restores every stream's state and either re-raises ``exc`` (default) or isolates the
    stream (``isolate_faults``)."""

    def __init__(self, stream: int, exc: Exception):
        super().__init__(f"stream {stream}: {exc}")
        self.stream, self.exc = int(stream), exc


_AR = np.arange(1 << 16, dtype=np.int64)
_PAT_CACHE = {}


def _ar(n: int) -> np.ndarray:
    """arange(n) as a view of a cached array (n is small on this path)."""
    return _AR[:n] if n <= _AR.shape[0] else np.arange(n, dtype=np.int64)


def _conv2_pattern(t2: int, F1: int, F2: int) -> np.ndarray:
    """Row offsets (into the channels-last conv1 output) of the top-left tap of
    every (t2, f2) output position of the second 3x3/stride-2 convolution."""
    key = (t2, F1, F2)
    pat = _PAT_CACHE.get(key)
    if pat is None:
        pat = ((2 * _ar(t2))[:, None] * F1 + 2 * _ar(F2)[None, :]).reshape(-1)
        _PAT_CACHE[key] = pat
    return pat


@dataclass
class StreamState:
    """Host mirror of one stream's scalar state (everything else is in HBM)."""
    # frontend (apply_frontend)
    fe_started: bool = False      # prev_states is not None
    pcm_start: int = 0            # first un-consumed sample in the device pcm buffer
    pcm_end: int = 0              # one past the last received sample
    # encoder (forward_infer)
    enc_started: bool = False     # prev_states is not None
    fpp: int = 0                  # ping-pong index of the pre-subsampling feature buffer
    nfeat: int = 0                # rows held in buffer_before_downsampling
    upp: int = 0
    nsub: int = 0                 # rows held in buffer_after_downsampling (-1: None)
    has_sub: bool = False
    n_blocks: int = 0             # n_processed_blocks
    has_addin: bool = False
    has_ctx: bool = False
    short_pos: int = 0            # StreamPositionalEncoding counter (never reset: A13)
    # search (BlockwiseSynchronousBeamSearch)
    T_enc: int = 0                # len(encoder_buffer)
    processed_block: int = 0
    process_idx: int = 0
    prev_valid: bool = False
    started: bool = False         # running_hyps initialised
    cur: int = 0                  # which ping-pong hypothesis buffer is live
    L: int = 1                    # tokens per live hypothesis (incl. sos)
    nhyp: int = 1
    has_ctc: bool = False         # live hypotheses carry a CTC state
    T_ctc: int = 0                # rows in the CTC table (strict_reference: survives reset(), see StreamBatch.reset)
    T_kv: int = 0                 # encoder frames already projected to cross-attention K|V rows
    output_index: int = 0
    n_steps_total: int = 0


class StreamBatch:
    """S independent streams sharing one weight replica on one GPU."""

    def __init__(self, weights: PackedWeights, backend, n_streams: int,
                 search: SearchConfig = SearchConfig(), max_frames: int = 1600,
                 max_tokens: int = 640, pcm_capacity: int = 1 << 20,
                 max_chunk_samples: int = 32768, strict_reference: bool = True, kv_dtype: str = "float32",
                 kv_pool_rows: int = 0):
        self.w = weights
        self.cfg = cfg = weights.cfg
        self.be = backend
        self.S = S = n_streams
        self.search = search
        self.W = W = search.beam_size
        self.K = K = min(search.pre_beam, cfg.vocab_size)
        if W > K:
            raise EngineError("beam_size must not exceed the pre-beam size (40)")
        if hasattr(backend, "check_supported"):
            backend.check_supported(cfg, W)
        self.TCAP, self.LCAP, self.PCAP = max_frames, max_tokens, pcm_capacity
        self.strict_reference = strict_reference
        dev = weights.device
        d, V, F = cfg.d_model, cfg.vocab_size, cfg.ffn_dim
        self.dev = dev
        f32, i32, f64 = torch.float32, torch.int32, torch.float64
        z = lambda *shape, dtype=f32: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731

        # ---- frontend / encoder buffers
        self.max_chunk_samples = max_chunk_samples
        self.max_feat_new = 2 + max_chunk_samples // cfg.hop_length + 8
        self.FCAP = self.max_feat_new + 16
        self.UCAP = cfg.block_size + self.FCAP // 4 + 8
        self.pcm = z(S, self.PCAP)
        self.featbuf = z(2 * S * self.FCAP, cfg.n_mels)
        self.subbuf = z(2 * S * self.UCAP, d)
        self.prev_addin = z(S, d)
        self.past_ctx = z(S * cfg.enc_layers, d)
        self.enc = z(S * self.TCAP, d)
        # scratch for one encoder call over all streams
        self.max_t1 = (self.FCAP - 3) // 2 + 1
        self.max_t2 = (self.max_t1 - 3) // 2 + 1
        self.c1 = z(S * self.max_t1 * cfg.conv_freq1, d)
        self.c2 = z(S * self.max_t2 * cfg.conv_freq2, d)
        self.max_blocks = S * (self.UCAP // cfg.hop_size + 1)
        R = cfg.block_size + 2
        m_enc = self.max_blocks * R
        self.xblk = z(m_enc, d)
        self.ws_xn = z(m_enc, d)
        self.ws_qkv = z(m_enc, 3 * d)
        self.ws_att = z(m_enc, d)
        self.ws_ffh = z(m_enc, F)
        # persistent job tables (stable addresses: the launch sequences that read
        # them are replayed from hipGraphs)
        self.jobs_ctx = z(S, 4, dtype=torch.int32)
        # ---- search state
        self.ctcx = z(S * self.TCAP, V)
        self.ctcxT = z(S * V, (self.TCAP + 3) // 4 * 4)     # column-major copy streamed by the prefix scan
        # K|V caches: fp32 (the reference's arithmetic) or fp16 storage (kv_dtype="float16": half the attention
        # kernels' HBM stream; arithmetic stays fp32; HIP backend only)
        kvt = {"float32": f32, "float16": torch.float16}[kv_dtype]
        if kvt != f32 and not hasattr(backend, "kv_rows_to_half"):
            raise EngineError("half-precision K|V caches need the HIP backend")
        self.ckv = z(S * cfg.dec_layers * self.TCAP, 2 * d, dtype=kvt)
        # self-attention K|V: a pool of rows per stream and layer (include/scasr.h: sc_search.skv); `anc` holds pool rows
        # default (csrc/streams.hip: sc_streams_create): one row per (position, hypothesis) - never exhausted before max_tokens -
        # unless that takes more than a quarter of the device's free memory; never fewer than 1.5 rows per position + 4W
        if kv_pool_rows <= 0:
            kv_pool_rows = self.LCAP * W
            if torch.device(dev).type == "cuda":
                per_row = S * cfg.dec_layers * 2 * d * (2 if kvt == torch.float16 else 4)
                kv_pool_rows = max(self.LCAP + self.LCAP // 2 + 4 * W,
                                   min(kv_pool_rows, torch.cuda.mem_get_info(torch.device(dev))[0] // 4 // per_row))
            kv_pool_rows = min(kv_pool_rows, 65536)
        self.kv_rows = max(2 * W, min(self.LCAP * W, kv_pool_rows))
        if self.kv_rows > 65536:
            raise EngineError("the self-attention K|V pool is limited to 65536 rows per stream (max_tokens / kv_pool_rows)")
        self.skv = z(S * cfg.dec_layers * self.kv_rows, 2 * d, dtype=kvt)
        self._flags2 = z(2 * S, dtype=i32)          # stop flags of the prune kernel | "K|V pool exhausted" of sc_kv_alloc
        self.kvflags = self._flags2[S:]
        self.kv_stage_rows = max(256, S * 24)
        self.kv_stage = z(cfg.dec_layers * self.kv_stage_rows, 2 * d) if kvt != f32 else None
        self.yseq = z(2, S, W, self.LCAP, dtype=i32)
        self.xpos = z(2, S, W, self.LCAP, dtype=i32)
        self.anc = z(2, S, self.LCAP, W, dtype=i32)
        self.score = z(2, S, W, dtype=f64)
        self.sc_dec = z(2, S, W, dtype=f64)
        self.sc_ctc = z(2, S, W, dtype=f64)
        self.ctc_r = z(2, S, self.TCAP, 2, W)
        self.ctc_rs = z(2, S, self.TCAP, W)      # log(exp r^n + exp r^b) of ctc_r per frame (scasr.h: sc_search.ctc_rs)
        self.ctc_s = z(2, S, W)
        self.ctc_rnew = z(S, (self.TCAP + 15) // 16, 2, W * K)     # checkpoints: r of the candidates at frames t % 16 == 15
        # ctrl rows and the compaction row map share one buffer: one upload per step
        self._ctrlmap = z(S * 8 + S * W, dtype=i32)
        self.ctrl = self._ctrlmap[: S * 8].view(S, 8)
        self.rowmap = self._ctrlmap[S * 8:]
        self.rowmap.copy_(torch.arange(S * W, dtype=i32))
        self.n_rows_step = S * W
        self._decode_prepared = False
        self._isolate, self._faults = False, {}
        self.flags = self._flags2[:S]
        # pinned host mirrors: the per-step ctrl upload / flag read-back are the
        # only host<->device traffic of the decode loop
        pin = dev.type == "cuda"
        self._ctrlmap_host = torch.zeros(S * 8 + S * W, dtype=i32, pin_memory=pin)
        self._ctrl_host = self._ctrlmap_host[: S * 8].view(S, 8)
        self._rowmap_np = self._ctrlmap_host[S * 8:].numpy()
        self._rowmap_np[:] = np.arange(S * W, dtype=np.int32)
        self._hyp_ofs = np.arange(W, dtype=np.int64)
        self._stream_ids = np.arange(S, dtype=np.int64)
        self._rowmap_key = None
        self._rowmap_identity = np.arange(S * W, dtype=np.int32)
        self.row_bucket = max(1, S // 16)        # compaction granularity in streams
        self._flags_host = torch.zeros(2 * S, dtype=i32, pin_memory=pin)
        self._ctrl_host0 = torch.zeros(S, 8, dtype=i32, pin_memory=pin)   # block-start upload (own buffer:
        self._ctrl_np0 = self._ctrl_host0.numpy()                         # it may still be in flight when step 1 is built)
        self._ctrl_np = self._ctrl_host.numpy()
        # staging arena for the per-call job / row tables
        self._arena_cap = (1 << 22) if pin else 0
        self._arena_off = 0
        if pin:
            self._arena_host = torch.zeros(self._arena_cap, dtype=i32, pin_memory=True)
            self._arena_np = self._arena_host.numpy()
            self._arena_dev = torch.zeros(self._arena_cap, dtype=i32, device=dev)
        self._flags_np = self._flags_host.numpy()
        n = S * W
        self.dx = z(n, d)
        self.dxn = z(n, d)
        self.dqkv = z(n, 3 * d)
        self.datt = z(n, d)
        self.dq = z(n, d)
        self.dffh = z(n, F)
        self.logits = z(n, V)
        self.logp = z(n, V)
        self.pre_ids = z(n, K, dtype=i32)
        self.psi = z(n, K)
        self.psi_eos = z(n)
        self.cand_score = z(n, W)
        self.cand_tok = z(n, W, dtype=i32)
        self.cand_ctc = z(n, W)
        self.sel = z(S, W, 2, dtype=i32)
        # head-parallel decoder layers (csrc/decoder_layer.hip): per-head partial products of the two attention
        # output projections and the feed-forward partial sums (by row id)
        from .weights import dec_layer_fused_supported
        if dec_layer_fused_supported(cfg, W):
            self.ph1 = z(n, cfg.dec_heads, d)
            self.ph2 = z(n, cfg.dec_heads, d)
            self.ffn_part = z(F // 128, n, d)
        else:
            self.ph1 = self.ph2 = self.ffn_part = None

        # all device work of this batch runs on one dedicated (non-default) HIP
        # stream, which also makes the decode step capturable as a hipGraph
        self.stream = torch.cuda.Stream(device=dev, priority=-1) if dev.type == "cuda" else None
        if self.stream is not None and hasattr(backend, "bind_stream"):
            backend.bind_stream(self.stream)
        self.st = [StreamState() for _ in range(S)]
        self.reset_all()
        self.stats = {"enc_calls": 0, "dec_steps": 0, "dec_blocks": 0}
        # optional host-side phase timers (SC_TIMING=1): seconds per phase
        self.timing = {} if os.environ.get("SC_TIMING") == "1" else None

    # ------------------------------------------------------------------
    def _tick(self, name, t0):
        if self.timing is not None:
            self.timing[name] = self.timing.get(name, 0.0) + (time.perf_counter() - t0)

    def _upload_ctrl(self):
        """ctrl rows -> device (async from pinned memory when on a GPU; the
        kernels that read them are ordered behind the copy on the same stream)."""
        self._ctrlmap.copy_(self._ctrlmap_host, non_blocking=self.stream is not None)

    def _set_rowmap(self, active_streams: np.ndarray):
        """Dense decoder kernels process the first n_rows_step entries of rowmap:
        the hypothesis rows of the streams still in the step loop, rounded up to
        a bucket of row_bucket streams (one captured graph per bucket) with rows
        of idle streams (their activations are scratch until their next step)."""
        S, W = self.S, self.W
        na = int(active_streams.size)
        nb = min(S, -(-na // self.row_bucket) * self.row_bucket)
        key = active_streams.tobytes()
        if key != self._rowmap_key:          # the active set changes only when a stream stops / starts
            self._rowmap_key = key
            if na == S:
                self._rowmap_np[:] = self._rowmap_identity
            else:
                idle = np.ones(S, bool)
                idle[active_streams] = False
                order = np.argsort(idle, kind="stable")      # active streams first, both parts in stream order
                np.add((order * W)[:, None], self._hyp_ofs[None, :], out=self._rowmap_np.reshape(S, W), casting="unsafe")
        self.n_rows_step = nb * W

    def _read_flags(self) -> np.ndarray:
        if self.stream is not None:
            self._flags_host.copy_(self._flags2, non_blocking=True)
            self.stream.synchronize()
        else:
            self._flags_host.copy_(self._flags2)
        return self._flags_np

    def _itensor(self, arr) -> torch.Tensor:
        """Host int table -> device int32 tensor.  On a GPU the table is staged
        in a pinned arena and copied asynchronously on the batch's stream (no
        host sync: launches that read it are ordered behind the copy); the
        arena is recycled at the start of every push, after the previous push's
        final synchronisation."""
        a = np.ascontiguousarray(arr, dtype=np.int32)
        n = a.size
        if self.stream is None or self._arena_off + n > self._arena_cap:
            return torch.from_numpy(a).to(self.dev, non_blocking=False)
        off = self._arena_off
        self._arena_off = off + ((n + 63) & ~63)
        self._arena_np[off:off + n] = a.reshape(-1)
        dst = self._arena_dev[off:off + n]
        dst.copy_(self._arena_host[off:off + n], non_blocking=True)
        return dst.view(a.shape)

    def reset(self, s: int):
        """Speech2TextStreaming.reset + BlockwiseSynchronousBeamSearch.reset
        (speech2text_streaming.py:252-263, beam_search.py:343-356).

        ``strict_reference`` (default) reproduces what the reference leaves behind:
        CTCPrefixScorer.impl is never cleared (scorers.py:342-350), so the next
        utterance on this stream is scored over the STALE CTC table - its rows
        [0, T_old) keep the previous utterance's posteriors and the table only
        grows once the new utterance has more than T_old frames (fixture
        tests/golden/tiny_reset.json) - and the StreamPositionalEncoding counter of
        the short-segment path keeps counting (A13).  With strict_reference=False
        the stream restarts from a clean state."""
        old = self.st[s]
        ns = StreamState()
        if self.strict_reference:
            ns.short_pos = old.short_pos  # A13: counter survives reset()
            ns.T_ctc = old.T_ctc          # stale CTC table (rows stay in self.ctcx)
        self.st[s] = ns
        if getattr(self, "stream", None) is not None:
            with torch.cuda.stream(self.stream):   # same stream as the kernels that read it
                self._init_hyp(s)
        else:
            self._init_hyp(s)

    def reset_all(self):
        for s in range(self.S):
            self.reset(s)

    def _init_hyp(self, s: int):
        # create_initial_hypothesis (hypothesis.py:75-91): yseq=[sos], xpos=[0]
        self.yseq[0, s, 0, 0] = self.cfg.sos_id
        self.xpos[0, s, 0, 0] = 0
        self.anc[0, s, 0, 0] = 0          # K|V pool row of the sos token
        self.score[0, s, 0] = 0.0
        self.sc_dec[0, s, 0] = 0.0
        self.sc_ctc[0, s, 0] = 0.0

    # ------------------------------------------------------------------
    # frontend planning  (apply_frontend)
    # ------------------------------------------------------------------
    def _plan_frontend(self, st: StreamState, is_final: bool):
        """Returns None (nothing to emit) or (seg_start, seg_len, eff_len,
        keep_lo, keep_n) and updates pcm_start / fe_started exactly like
        apply_frontend (speech2text_streaming.py:300-400)."""
        cfg = self.cfg
        win, hop = cfg.win_length, cfg.hop_length
        N = st.pcm_end - st.pcm_start
        first = not st.fe_started
        seg_start = st.pcm_start
        trim = math.ceil(math.ceil(win / hop) / 2)
        if not N > win and not is_final:
            st.fe_started = True          # next_states = {"waveform_buffer": speech}
            return None
        if is_final:
            eff = N if N > win else win   # zero-pad up to win_length
            total = 1 + eff // hop
            lo, n = 0, total
            if not first and total > trim:
                lo, n = trim, total - trim
            st.fe_started = False         # next_states = None
            st.pcm_start = st.pcm_end
            return seg_start, N, eff, lo, n
        n_frames = (N - (win - hop)) // hop
        n_res = (N - (win - hop)) % hop
        proc = (win - hop) + n_frames * hop
        total = 1 + proc // hop
        st.pcm_start = st.pcm_end - (win - hop) - n_res
        st.fe_started = True
        if first:
            n = total - trim if total > trim else total
            return seg_start, proc, proc, 0, n
        if total > 2 * trim:
            return seg_start, proc, proc, trim, total - 2 * trim
        return None                       # "too short after trimming": frames are lost

    # ------------------------------------------------------------------
    def push(self, chunks: Sequence[Tuple[int, Optional[np.ndarray], bool]],
             pcm_resident: bool = False, isolate_faults: bool = False):
        """One chunk step, run to completion (every block finishes inside the call: the reference's per-call
        semantics).  Returns {stream: has_output}.  A stream whose chunk cannot be processed (capacity limit;
        a final chunk with fewer than 7 feature frames, on which the reference raises too - A3)
        raises its exception with every stream's state as it was before the call; with
        ``isolate_faults`` that stream alone is reset and its entry in the result is the exception
        object - the reference's "one failing websocket does not disturb the others"
        (speechcatcher_server.py:359-397, one model instance per client)."""
        self._isolate, self._faults = bool(isolate_faults), {}
        if self.stream is None:
            out, feat_new, finals = self._stage_encode_safe(chunks, pcm_resident)
            self._stage_decode(feat_new, finals, {s: self.st[s].T_enc for s in feat_new})
            return self._finish_faults(out)
        self._arena_off = 0
        with torch.cuda.stream(self.stream):
            out, feat_new, finals = self._stage_encode_safe(chunks, pcm_resident)
            self._stage_decode(feat_new, finals, {s: self.st[s].T_enc for s in feat_new})
        self.stream.synchronize()
        return self._finish_faults(out)

    def _finish_faults(self, out):
        """isolate_faults: a stream that failed (capacity limit, or an input the reference itself
        dies on) is reset and reports its exception in place of the has-output flag; every other
        stream of the step is unaffected."""
        for s, exc in self._faults.items():
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    self.reset(s)
            else:
                self.reset(s)
            out[s] = exc
        self._faults = {}
        return out

    def _stage_encode_safe(self, chunks, pcm_resident):
        """_stage_encode with per-stream fault handling.  All planning of a chunk step is pure host
        integer work that runs BEFORE the first launch that changes device state, so a failure is
        undone by restoring the host mirrors: every stream is back in the state it had before the
        call.  Then the fault is either raised (default) or the offending stream is dropped from
        the step (``isolate_faults``) and the step is planned again for the others."""
        chunks = list(chunks)
        while True:
            for s, samples, _fin in chunks:   # compaction moves device data: settle it before the snapshot
                n_new = int(samples) if pcm_resident else int(len(samples))
                st = self.st[s]
                if st.pcm_end + n_new > self.PCAP:
                    self._compact_pcm(s)
            snap = {s: copy.copy(self.st[s]) for s, _, _ in chunks}
            try:
                return self._stage_encode(chunks, pcm_resident)
            except StreamFault as f:
                for s, st in snap.items():
                    self.st[s] = st
                if not self._isolate:
                    raise f.exc
                self._faults[f.stream] = f.exc
                chunks = [c for c in chunks if c[0] != f.stream]
                if not chunks:
                    return {}, {}, {}

    def _stage_encode(self, chunks, pcm_resident=False):
        """Frontend + encoder of one chunk step for the listed streams: (stream, samples, is_final).

        ``samples`` is 1-D float PCM in +-1 (np.ndarray / torch tensor); with
        ``pcm_resident`` the samples are already in ``self.pcm`` and the
        tuple carries the sample COUNT instead (bench path: inputs resident
        in HBM).  Mirrors Speech2TextStreaming.__call__ for raw audio.
        Returns {stream: has_output} where has_output=False reproduces the
        reference's early ``return []`` (speech2text_streaming.py:432-433).
        """
        cfg = self.cfg
        t_ph = time.perf_counter()
        fe_jobs = []
        feat_new: Dict[int, int] = {}
        finals: Dict[int, bool] = {}
        for s, samples, is_final in chunks:
            st = self.st[s]
            finals[s] = bool(is_final)
            n_new = int(samples) if pcm_resident else int(len(samples))
            if st.pcm_end + n_new > self.PCAP:
                raise StreamFault(s, EngineError(
                    f"pcm buffer capacity exceeded (pcm_capacity={self.PCAP} samples per stream)"))
            if not pcm_resident and n_new > 0:
                t = torch.as_tensor(samples, dtype=torch.float32)
                self.pcm[s, st.pcm_end: st.pcm_end + n_new].copy_(t, non_blocking=False)
            st.pcm_end += n_new
            plan = self._plan_frontend(st, is_final)
            if plan is None:
                continue
            seg_start, seg_len, eff_len, lo, n = plan
            if n > self.max_feat_new:
                raise StreamFault(s, EngineError(
                    f"a call of {n_new} samples produces {n} feature frames; this batch was built for at most "
                    f"{self.max_feat_new} (max_chunk_samples={self.max_chunk_samples})"))
            # encoder buffer_before_downsampling lives at rows [0, nfeat) of the
            # live ping-pong half; new frames are appended behind it.
            nbuf = st.nfeat if st.enc_started else 0
            dst_row0 = (st.fpp * self.S + s) * self.FCAP + nbuf
            fe_jobs.append((s, seg_start, seg_len, eff_len, lo, n, dst_row0, 0))
            feat_new[s] = n
        if fe_jobs:
            jobs = self._itensor(np.array(fe_jobs, np.int32))
            self.be.logmel(self.w, self.pcm, self.PCAP, jobs, len(fe_jobs),
                           max(j[5] for j in fe_jobs), self.featbuf)
        out = {s: (s in feat_new) for s, _, _ in chunks}
        self._tick("frontend_host", t_ph)
        if feat_new:
            self._encode_features(feat_new, finals)
        return out, feat_new, finals

    def push_features(self, items: Sequence[Tuple[int, torch.Tensor, bool]], isolate_faults: bool = False):
        self._isolate, self._faults = bool(isolate_faults), {}
        if self.stream is None:
            return self._finish_faults(self._push_features(items))
        self._arena_off = 0
        with torch.cuda.stream(self.stream):
            out = self._push_features(items)
        self.stream.synchronize()
        return self._finish_faults(out)

    def _push_features(self, items):
        """2-D (T, n_mels) already-normalised features (the reference's 2-D /
        3-D input path, speech2text_streaming.py:438-449).  Capacity is checked for
        every item BEFORE anything is copied: a feature matrix must fit the stream's
        slot of the feature buffer (rows held + new <= FCAP, new <= max_feat_new - the
        conv scratch is sized for that), else rows would spill into the next stream's slot."""
        items = list(items)
        while True:
            snap = {s: copy.copy(self.st[s]) for s, _, _ in items}
            try:
                feat_new, finals = {}, {}
                for s, feats, is_final in items:
                    st = self.st[s]
                    n = int(feats.shape[0])
                    nbuf = st.nfeat if st.enc_started else 0
                    if feats.ndim != 2 or feats.shape[1] != self.cfg.n_mels:
                        raise StreamFault(s, EngineError(f"features must be (T, {self.cfg.n_mels})"))
                    if n > self.max_feat_new or nbuf + n > self.FCAP:
                        raise StreamFault(s, EngineError(
                            f"{n} feature frames in one call (+{nbuf} buffered) exceed this batch's capacity of "
                            f"{self.max_feat_new} new frames per call (max_chunk_samples={self.max_chunk_samples})"))
                for s, feats, is_final in items:
                    st = self.st[s]
                    n = int(feats.shape[0])
                    nbuf = st.nfeat if st.enc_started else 0
                    r0 = (st.fpp * self.S + s) * self.FCAP + nbuf
                    self.featbuf[r0:r0 + n].copy_(torch.as_tensor(feats, dtype=torch.float32))   # appended rows only
                    feat_new[s] = n
                    finals[s] = bool(is_final)
                self._encode_features(feat_new, finals)
                break
            except StreamFault as f:
                for s, st in snap.items():
                    self.st[s] = st
                if not self._isolate:
                    raise f.exc
                self._faults[f.stream] = f.exc
                items = [it for it in items if it[0] != f.stream]
                if not items:
                    return {}
        self._stage_decode(feat_new, finals, {s: self.st[s].T_enc for s in feat_new})
        return {s: True for s in feat_new}

    def _compact_pcm(self, s: int):
        st = self.st[s]
        n = st.pcm_end - st.pcm_start
        if st.pcm_start > 0:
            self.pcm[s, :n] = self.pcm[s, st.pcm_start:st.pcm_end].clone()
            st.pcm_start, st.pcm_end = 0, n

    # ------------------------------------------------------------------
    # process_block: encoder + decode schedule
    # ------------------------------------------------------------------
    def _encode_features(self, feat_new: Dict[int, int], finals: Dict[int, bool]):
        for s in feat_new:
            st = self.st[s]
            if not st.started:
                st.started = True  # running_hyps = [initial hypothesis]
        enc_streams = [s for s, n in feat_new.items() if n >= 3]
        # n < 3: encoder skipped, frames discarded (beam_search.py:551-559)
        if enc_streams:
            t_ph = time.perf_counter()
            self._encode(enc_streams, feat_new, finals)
            self._tick("encode_host", t_ph)

    def _stage_decode(self, feat_new: Dict[int, int], finals: Dict[int, bool], t_avail: Dict[int, int]):
        """Decode schedule of one chunk step.  ``t_avail[s]`` = encoder frames of
        stream s after this step's encoder stage."""
        cfg = self.cfg
        if not feat_new:
            return
        # decode schedule (beam_search.py:590-634), rounds of lock-step blocks
        pending = list(feat_new.keys())
        done_final = set()
        while True:
            todo = []
            for s in pending:
                st = self.st[s]
                if s in done_final:
                    continue
                cur_end = cfg.block_size - cfg.look_ahead + cfg.hop_size * st.processed_block
                if t_avail[s] > 0 and cur_end < t_avail[s]:
                    todo.append((s, cur_end, False))
                elif finals[s] and t_avail[s] > 0:
                    todo.append((s, t_avail[s], True))
                    done_final.add(s)
            if not todo:
                break
            t_ph = time.perf_counter()
            self._decode_blocks(todo)
            self._tick("decode_total", t_ph)
            for s, _, fin in todo:
                if not fin:
                    self.st[s].processed_block += 1

    # ------------------------------------------------------------------
    _ENC_STATE_FIELDS = ("enc_started", "nfeat", "fpp", "has_sub", "nsub", "upp", "n_blocks",
                         "has_addin", "has_ctx", "T_enc")

    def _encode(self, streams: List[int], feat_new: Dict[int, int], finals: Dict[int, bool]):
        """forward_infer for every listed stream, batched.

        Planning is pure integer work.  When every stream of the call is in the
        same buffering state (lock-step batches: the serving case) the plan is
        computed for ONE stream and broadcast with per-stream offsets, so the
        host cost does not grow with the number of streams."""
        st0 = self.st[streams[0]]
        sig0 = (finals[streams[0]], feat_new[streams[0]]) + tuple(getattr(st0, f) for f in self._ENC_STATE_FIELDS)
        uniform = len(streams) > 2 and all(
            (finals[s], feat_new[s]) + tuple(getattr(self.st[s], f) for f in self._ENC_STATE_FIELDS) == sig0
            for s in streams[1:])
        if uniform:
            plan = self._encode_plan(streams[:1], feat_new, finals)
            if plan is not None and not plan["short_jobs"]:
                plan = self._broadcast_plan(plan, streams)
                for s in streams[1:]:
                    for f in self._ENC_STATE_FIELDS:
                        setattr(self.st[s], f, getattr(st0, f))
            elif plan is not None:
                # short-segment path: plan the remaining streams the general way
                rest = self._encode_plan(streams[1:], feat_new, finals)
                self._encode_launch(plan)
                plan = rest
            else:
                for s in streams[1:]:
                    for f in self._ENC_STATE_FIELDS:
                        setattr(self.st[s], f, getattr(st0, f))
        else:
            plan = self._encode_plan(streams, feat_new, finals)
        if plan is not None:
            self._encode_launch(plan)

    def _broadcast_plan(self, p, streams):
        """Replicate the single-stream plan of streams[0] to all streams."""
        cfg = self.cfg
        n = len(streams)
        sv = np.asarray(streams, np.int64)
        ds = sv - sv[0]                      # stream-id distance
        j = np.arange(n, dtype=np.int64)     # dense position in this call
        R = cfg.block_size + 2
        F1 = cfg.conv_freq1
        q = dict(p)
        cj = np.asarray(p["conv_jobs"], np.int64)            # (1, 4)
        t1 = int(cj[0, 3])
        cjn = np.repeat(cj, n, 0)
        cjn[:, 0] += ds * self.FCAP
        cjn[:, 2] += j * t1
        q["conv_jobs"] = cjn
        q["a_rows"] = (p["a_rows"][None, :] + (j * t1 * F1)[:, None]).reshape(-1)
        q["lin_dst"] = (p["lin_dst"][None, :] + (ds * self.UCAP)[:, None]).reshape(-1)
        for key, stride in (("feat_copy", self.FCAP), ("sub_copy", self.UCAP)):
            if p[key] is not None:
                q[key] = tuple((a[None, :] + (ds * stride)[:, None]).reshape(-1) for a in p[key])
        if p["blk_jobs"] is not None:
            bj = p["blk_jobs"]                               # (nb, 6)
            nb = bj.shape[0]
            bjn = np.tile(bj, (n, 1))
            bjn[:, 0] += np.repeat(ds * self.UCAP, nb)
            q["blk_jobs"] = bjn
            sj = np.repeat(p["sjobs"], n, 0)                 # (n, 5)
            sj[:, 0] += j * nb
            sj[:, 2] = sv
            q["sjobs"] = sj
        else:
            nb = 0
        if p["emit_src"] is not None:
            es = p["emit_src"]
            esn = np.tile(es, n)
            add = np.repeat(j * nb * R, es.shape[0])
            q["emit_src"] = np.where(esn >= 0, esn + add, esn)
            q["emit_dst"] = (p["emit_dst"][None, :] + (ds * self.TCAP)[:, None]).reshape(-1)
        return q

    def _encode_plan(self, streams, feat_new, finals):
        cfg, S = self.cfg, self.S
        F1, F2 = cfg.conv_freq1, cfg.conv_freq2
        sub = cfg.subsample
        conv_jobs = []      # (src_row0, T_in, c1_row0, T1)
        a_rows = []         # conv2 implicit-GEMM A row indices into c1 (rows of width d)
        lin_dst = []        # dst rows in subbuf for the subsampling Linear output
        feat_copy = ([], [])
        c1_rows = 0
        per = {}
        # ---- stage 1: buffer_before_downsampling + Conv2dSubsampling (:278-311)
        for s in streams:
            st = self.st[s]
            fin = finals[s]
            nbuf = st.nfeat if st.enc_started else 0
            Tf = nbuf + feat_new[s]
            base = (st.fpp * S + s) * self.FCAP
            st.enc_started = True
            if fin:
                t_use, keep = Tf, 0
                if Tf < 7:
                    # the reference dies inside Conv2d with RuntimeError (A3)
                    raise StreamFault(s, RuntimeError(
                        "Calculated padded input size per channel is smaller than the 3x3 "
                        f"subsampling kernel (stream {s}: {Tf} feature frames in a final chunk)"))
            else:
                n_s = Tf // sub - 1
                if n_s < 2:
                    st.nfeat = Tf
                    continue
                keep = Tf % sub + sub * 2
                t_use = n_s * sub
            t1 = (t_use - 3) // 2 + 1
            t2 = (t1 - 3) // 2 + 1
            conv_jobs.append((base, t_use, c1_rows, t1))
            a_rows.append(c1_rows * F1 + _conv2_pattern(t2, F1, F2))
            c1_rows += t1
            if keep:
                # residual feature rows move to the other ping-pong half
                obase = ((1 - st.fpp) * S + s) * self.FCAP
                feat_copy[0].append(base + Tf - keep + _ar(keep))
                feat_copy[1].append(obase + _ar(keep))
                st.fpp = 1 - st.fpp
            st.nfeat = keep
            nsub = st.nsub if st.has_sub else 0
            ubase = (st.upp * S + s) * self.UCAP
            if nsub + t2 > self.UCAP:
                raise StreamFault(s, EngineError("subsampled-frame buffer capacity exceeded"))
            lin_dst.append(ubase + nsub + _ar(t2))
            per[s] = (t2, nsub, ubase)
        if not conv_jobs:
            return None
        # ---- stage 2: buffer_after_downsampling, block extraction (:313-380)
        R = cfg.block_size + 2
        offset = cfg.block_size - cfg.look_ahead - cfg.hop_size
        blk_jobs = []          # (src_row0, chunk_len, pe_off_frames, pe_off_ctx, short, 0)
        sjobs = []             # (b0, nblk, stream, has_addin, has_ctx)
        emit_src, emit_dst = [], []
        sub_copy = ([], [])
        short_jobs = []
        for s in streams:
            if s not in per:
                continue
            st = self.st[s]
            fin = finals[s]
            t2, nsub, ubase = per[s]
            U = nsub + t2
            if fin:
                nb = math.ceil(float(U - offset - cfg.look_ahead) / float(cfg.hop_size))
                if st.n_blocks == 0 and U <= cfg.block_size:
                    if st.T_enc + U > self.TCAP:
                        raise StreamFault(s, EngineError(f"encoder-frame capacity exceeded (max_frames={self.TCAP})"))
                    short_jobs.append((s, ubase, U))
                    continue
            else:
                if U <= cfg.block_size:
                    st.has_sub, st.nsub = True, U
                    continue
                overlap = cfg.block_size - cfg.hop_size
                nb = max(0, U - overlap) // cfg.hop_size
                res = U - cfg.hop_size * nb
                obase = ((1 - st.upp) * S + s) * self.UCAP
                sub_copy[0].append(ubase + U - res + _ar(res))
                sub_copy[1].append(obase + _ar(res))
                st.upp = 1 - st.upp
                st.has_sub, st.nsub = True, res
            nb = max(nb, 0)
            b0 = len(blk_jobs)
            for i in range(nb):
                cur_hop = i * cfg.hop_size
                clen = min(cfg.block_size, U - cur_hop)
                blk_jobs.append((ubase + cur_hop, clen, cur_hop + cfg.hop_size * st.n_blocks,
                                 i + st.n_blocks, 0, 0))
            if nb > 0:
                sjobs.append((b0, nb, s, int(st.has_addin), int(st.has_ctx)))
                st.has_addin = st.has_ctx = True
            # output extraction (_extract_output_from_blocks_infer :500-522)
            first = st.n_blocks == 0
            if fin:
                y_len = U if first else U - offset
            else:
                y_len = nb * cfg.hop_size + (offset if first else 0)
            src = np.full(y_len, -1, np.int64)
            if first and nb > 0:
                src[0:offset] = (b0 * R) + 1 + _ar(offset)
            for i in range(nb):
                cur_hop = i * cfg.hop_size + (offset if first else 0)
                if i == nb - 1 and fin:
                    clen = min(cfg.block_size - offset, y_len - cur_hop)
                else:
                    clen = cfg.hop_size
                src[cur_hop:cur_hop + clen] = (b0 + i) * R + 1 + offset + _ar(clen)
            if st.T_enc + y_len > self.TCAP:
                raise StreamFault(s, EngineError(f"encoder-frame capacity exceeded (max_frames={self.TCAP})"))
            emit_src.append(src)
            emit_dst.append(s * self.TCAP + st.T_enc + _ar(y_len))
            st.T_enc += y_len
            st.n_blocks += nb
        # final call: next_states = None (:407-408)
        for s in streams:
            if finals[s]:
                st = self.st[s]
                st.enc_started = False
                st.nfeat = st.nsub = st.n_blocks = 0
                st.has_sub = st.has_addin = st.has_ctx = False
        cat = lambda lst: np.concatenate(lst) if lst else None  # noqa: E731
        return {
            "conv_jobs": np.asarray(conv_jobs, np.int64),
            "a_rows": np.concatenate(a_rows), "lin_dst": np.concatenate(lin_dst),
            "feat_copy": (cat(feat_copy[0]), cat(feat_copy[1])) if feat_copy[0] else None,
            "blk_jobs": np.asarray(blk_jobs, np.int64) if blk_jobs else None,
            "sjobs": np.asarray(sjobs, np.int64) if sjobs else None,
            "emit_src": cat(emit_src), "emit_dst": cat(emit_dst),
            "sub_copy": (cat(sub_copy[0]), cat(sub_copy[1])) if sub_copy[0] else None,
            "short_jobs": short_jobs,
        }

    def _encode_launch(self, p):
        cfg, be, w = self.cfg, self.be, self.w
        d, F1, F2 = cfg.d_model, cfg.conv_freq1, cfg.conv_freq2
        R = cfg.block_size + 2
        self.stats["enc_calls"] += 1
        cj = p["conv_jobs"]
        be.conv1(w, self.featbuf, self._itensor(cj), int(cj.shape[0]), int(cj[:, 3].max()), self.c1)
        a_rows = p["a_rows"]
        be.gemm(self.c1, self._itensor(a_rows), d, w.conv2_w, w.conv2_b, self.c2, None, d,
                int(a_rows.shape[0]), d, 9 * d, relu=True, conv_f1=F1)
        lin_dst = p["lin_dst"]
        be.gemm(self.c2, None, F2 * d, w.sub_out_w, w.sub_out_b, self.subbuf, self._itensor(lin_dst), d,
                int(lin_dst.shape[0]), d, F2 * d)
        if p["feat_copy"] is not None:
            src, dst = p["feat_copy"]
            be.copy_rows(self.featbuf, self._itensor(src), self.featbuf, self._itensor(dst),
                         int(src.size), cfg.n_mels)
        if p["blk_jobs"] is not None:
            bj, sj = p["blk_jobs"], p["sjobs"]
            nbk = int(bj.shape[0])
            if nbk > self.max_blocks:
                raise EngineError("too many encoder blocks in one call")
            be.block_pack(w, self.subbuf, self._itensor(bj), nbk, R, self.xblk)
            # slot-0 chain from prev_addin (:370-380)
            j_add = np.stack([sj[:, 0], sj[:, 1], sj[:, 2], sj[:, 3]], 1)
            be.ctx_handoff(self.xblk, R, self._itensor(j_add), int(sj.shape[0]), self.prev_addin, 0)
            j_ctx = np.stack([sj[:, 0], sj[:, 1], sj[:, 2] * cfg.enc_layers, sj[:, 4]], 1).astype(np.int32)
            ns = int(sj.shape[0])
            self.jobs_ctx[:ns].copy_(torch.from_numpy(np.ascontiguousarray(j_ctx)))
            be.encoder_layers(w, self.xblk, nbk, R, True, self.jobs_ctx, ns, self.past_ctx,
                              self.ws_xn, self.ws_qkv, self.ws_att, self.ws_ffh)
        if p["emit_src"] is not None and p["emit_src"].size:
            be.layernorm(self.xblk, self._itensor(p["emit_src"]), self.enc, self._itensor(p["emit_dst"]),
                         int(p["emit_src"].size), w.enc_norm_g, w.enc_norm_b)
        for s, ubase, U in p["short_jobs"]:
            self._encode_short(s, ubase, U)
        if p["sub_copy"] is not None:
            src, dst = p["sub_copy"]
            be.copy_rows(self.subbuf, self._itensor(src), self.subbuf, self._itensor(dst), int(src.size), d)

    def _encode_short(self, s: int, ubase: int, U: int):
        """Short-segment path (:345-351): one un-blocked pass, no mask, no
        context slots, StreamPositionalEncoding's internal counter (A13)."""
        cfg, w, be = self.cfg, self.w, self.be
        st = self.st[s]
        d = cfg.d_model
        job = np.array([[ubase, U, st.short_pos, 0, 1, 0]], np.int32)
        st.short_pos += U
        be.block_pack(w, self.subbuf, self._itensor(job), 1, U, self.xblk)
        be.encoder_layers(w, self.xblk, 1, U, False, None, 0, self.past_ctx,
                          self.ws_xn, self.ws_qkv, self.ws_att, self.ws_ffh)
        dst = s * self.TCAP + st.T_enc + _ar(U)
        be.layernorm(self.xblk, self._itensor(_ar(U)), self.enc, self._itensor(dst), U,
                     w.enc_norm_g, w.enc_norm_b)
        st.T_enc += U

    # ------------------------------------------------------------------
    # _decode_one_block for a lock-step group of streams
    # ------------------------------------------------------------------
    def _project_cross_kv(self, ar, kvt, m):
        """Cross-attention K|V rows of m new encoder frames, all decoder layers (projected ONCE per frame and
        shared by every hypothesis and step: decoder_layer.py:106-115 recomputes them per call)."""
        cfg, be, w = self.cfg, self.be, self.w
        d, Ld = cfg.d_model, cfg.dec_layers
        if self.kv_stage is None:
            for li in range(Ld):
                be.gemm(self.enc, ar, d, w.dec[li]["wkv"], w.dec[li]["bkv"], self.ckv[li * self.TCAP:],
                        kvt, 2 * d, m, 2 * d, d)
            return
        for r0 in range(0, m, self.kv_stage_rows):      # fp16 cache: fp32 staging, then convert + scatter all layers
            mm = min(self.kv_stage_rows, m - r0)
            for li in range(Ld):
                be.gemm(self.enc, ar[r0:], d, w.dec[li]["wkv"], w.dec[li]["bkv"], self.kv_stage[li * mm:], None, 2 * d,
                        mm, 2 * d, d)
            be.kv_rows_to_half(self.kv_stage, kvt[r0:], mm, Ld, self.TCAP, 2 * d, self.ckv)

    def _decode_blocks(self, todo: List[Tuple[int, int, bool]]):
        """_decode_one_block (beam_search.py:655-838) for a lock-step group of
        streams.  Per-stream scalars are handled as numpy vectors so the host
        cost per decode step is O(1) Python operations, not O(S)."""
        cfg, be, w = self.cfg, self.be, self.w
        S, W = self.S, self.W
        d, Ld = cfg.d_model, cfg.dec_layers
        n = len(todo)
        self.stats["dec_blocks"] += n
        ids = np.fromiter((t[0] for t in todo), dtype=np.int64, count=n)
        T = np.fromiter((t[1] for t in todo), dtype=np.int64, count=n)
        fin = np.fromiter((t[2] for t in todo), dtype=bool, count=n)
        sts = [self.st[s] for s in ids]
        cur = np.fromiter((x.cur for x in sts), dtype=np.int64, count=n)
        L = np.fromiter((x.L for x in sts), dtype=np.int64, count=n)
        nhyp = np.fromiter((x.nhyp for x in sts), dtype=np.int64, count=n)
        has = np.fromiter((x.has_ctc for x in sts), dtype=bool, count=n)
        pidx = np.fromiter((x.process_idx for x in sts), dtype=np.int64, count=n)
        pvalid = np.fromiter((x.prev_valid for x in sts), dtype=bool, count=n)
        told = np.fromiter((x.T_ctc for x in sts), dtype=np.int64, count=n)
        tkv = np.fromiter((x.T_kv for x in sts), dtype=np.int64, count=n)
        if (T > self.TCAP).any():
            raise EngineError("max_frames exceeded")
        # CTC table length after this block: the table never shrinks (a stale table left by reset()
        # stays longer than the new utterance until the utterance outgrows it, scorers.py:342-350)
        Ttab = np.maximum(T, told)
        # ---- extend_scorers (:403-464): CTC rows, cross-attention K/V rows, r states
        grow = T > told
        same_rows = bool((told == tkv).all())
        ar = None
        if grow.any():
            rows = np.concatenate([ids[i] * self.TCAP + _AR[told[i]:T[i]] for i in np.nonzero(grow)[0]])
            first = grow & (told == 0)
            ar = self._itensor(rows)
            m = int(rows.shape[0])
            be.gemm(self.enc, ar, d, w.ctc_w, w.ctc_b, self.ctcx, ar, cfg.vocab_size, m, cfg.vocab_size, d)
            if first.any():   # quirk A1: only the first block is log-softmaxed
                lr = np.concatenate([ids[i] * self.TCAP + _AR[0:T[i]] for i in np.nonzero(first)[0]])
                be.log_softmax_rows(self.ctcx, self._itensor(lr), int(lr.size), cfg.vocab_size)
        growkv = T > tkv
        if growkv.any():
            gi = np.nonzero(growkv)[0]
            kv0 = np.concatenate([ids[i] * Ld * self.TCAP + _AR[tkv[i]:T[i]] for i in gi])
            if not same_rows or ar is None:
                rows = np.concatenate([ids[i] * self.TCAP + _AR[tkv[i]:T[i]] for i in gi])
                ar = self._itensor(rows)
            m = int(kv0.shape[0])
            kvt = self._itensor(kv0)   # one row table for all layers: layer li's rows start li*TCAP rows further
            self._project_cross_kv(ar, kvt, m)
        if not self._decode_prepared:
            # one-time: capture the per-bucket decode graphs while every stream is idle on the device
            self._decode_prepared = True
            if hasattr(be, "prepare_decode"):
                self._ctrl_np[:] = 0
                self._upload_ctrl()
                be.prepare_decode(self)
        ctrl0 = self._ctrl_np0             # (every earlier use was followed by a flag read-back sync)
        ctrl0[:] = 0
        ctrl0[ids] = np.stack([np.ones(n, np.int64), cur, fin, Ttab, L, nhyp, has, told], 1)
        self.ctrl.copy_(self._ctrl_host0, non_blocking=self.stream is not None)
        be.ctc_extend_state(self)
        ctrl = self._ctrl_np
        ctrl[:] = 0
        for i, x in enumerate(sts):
            x.T_ctc = int(Ttab[i])
            x.T_kv = int(max(T[i], tkv[i]))
            x.output_index = 0
        # ---- step loop (:701-821)
        live = np.ones(n, bool)
        took_out = np.zeros(n, bool)     # loop left with a non-accepted H_out live
        nhyp_prev = nhyp.copy()
        has_prev = has.copy()
        out_idx = np.zeros(n, np.int64)
        nsteps = np.zeros(n, np.int64)
        use_bbd = self.search.use_bbd
        while True:
            act = live & (pidx < self.search.max_length)
            live &= act
            if not act.any():
                break
            over = act & (L + 1 > self.LCAP)
            if over.any():
                if not self._isolate:
                    raise EngineError(f"max_tokens exceeded (max_tokens={self.LCAP})")
                for i in np.nonzero(over)[0]:   # isolate: the stream leaves the loop here and is reset by push()
                    self._faults[int(ids[i])] = EngineError(f"max_tokens exceeded (max_tokens={self.LCAP})")
                live &= ~over
                act = act & ~over
                if not act.any():
                    break
            ctrl[ids] = np.stack([act, cur, fin, T, L, nhyp, has, Ttab], 1)
            self._set_rowmap(ids[act])
            self._upload_ctrl()
            self.stats["dec_steps"] += 1
            if "xattn_rows" in self.stats:   # bench.py roofline leg: K|V rows the cross-attention reads this step
                self.stats["xattn_rows"] += int(T[act].sum()) * Ld
            t_st = time.perf_counter()
            be.decode_step(self)
            self._tick("decode_launch", t_st)
            t_st = time.perf_counter()
            f = self._read_flags()[ids]
            self._tick("decode_wait_flags", t_st)
            kv_full = act & (self._flags_np[self.S:][ids] != 0)    # the prune kernel found no free K|V pool row
            f_any, f_best, f_all, f_rep = (f & F_ANY_EOS) != 0, (f & F_BEST_EOS) != 0, (f & F_ALL_EOS) != 0, (f & F_REPEAT) != 0
            stop_eos = act & f_any & (~fin | f_best)
            stop_bbd = act & ~stop_eos & f_rep & ~fin if use_bbd else np.zeros(n, bool)
            stop_all = act & ~stop_eos & ~stop_bbd & f_all & fin
            accept = act & ~(stop_eos | stop_bbd | stop_all)
            take = stop_eos | stop_all | accept
            # ... which only matters if the step is taken and stays taken: a step that is rolled back (stop_bbd) or rewound
            # when the block closes (not accepted, pidx > 1, prev_hyps valid) leaves the current side, which is fully valid
            full = kv_full & take & ~(~accept & (pidx > 1) & pvalid)
            if full.any():
                err = EngineError(f"self-attention K|V pool exhausted (kv_pool_rows={self.kv_rows}, max_tokens={self.LCAP})")
                if not self._isolate:
                    raise err
                for i in np.nonzero(full)[0]:   # isolate: the stream leaves the loop here and is reset by push()
                    self._faults[int(ids[i])] = err
                live &= ~full
                act = act & ~full
                stop_eos, stop_bbd, stop_all, accept, take = (v & ~full for v in (stop_eos, stop_bbd, stop_all, accept, take))
            out_idx += act
            nsteps += act
            out_idx -= stop_bbd
            nh_out = np.minimum(W, nhyp * W)
            nhyp_prev = np.where(take, nhyp, nhyp_prev)
            has_prev = np.where(take, has, has_prev)
            cur = np.where(take, 1 - cur, cur)
            L = L + take
            nhyp = np.where(take, nh_out, nhyp)
            has = has | take
            took_out |= stop_eos | stop_all
            live &= accept
            pvalid |= accept          # prev_hyps = copy(H_out)
            pidx += accept            # process_idx += 1
        # ---- rewind (:827-836)
        rw = (pidx > 1) & pvalid
        r2 = rw & took_out            # live state is a non-accepted H_out: go back to its H_in
        cur = np.where(r2, 1 - cur, cur)
        L = L - r2
        nhyp = np.where(r2, nhyp_prev, nhyp)
        has = np.where(r2, has_prev, has)
        pidx -= rw
        pvalid &= ~rw
        for i, x in enumerate(sts):
            x.cur, x.L, x.nhyp, x.has_ctc = int(cur[i]), int(L[i]), int(nhyp[i]), bool(has[i])
            x.process_idx, x.prev_valid = int(pidx[i]), bool(pvalid[i])
            x.output_index = int(out_idx[i])
            x.n_steps_total += int(nsteps[i])

    # ------------------------------------------------------------------
    def hypotheses(self, s: int):
        """Live hypotheses of stream s: list of dicts (yseq, score, scores, xpos)."""
        st = self.st[s]
        if not st.started:
            return []
        c, n, L = st.cur, st.nhyp, st.L
        ys = self.yseq[c, s, :n, :L].cpu().numpy()
        xp = self.xpos[c, s, :n, :L].cpu().numpy()
        sc = self.score[c, s, :n].cpu().numpy()
        sd = self.sc_dec[c, s, :n].cpu().numpy()
        scc = self.sc_ctc[c, s, :n].cpu().numpy()
        return [{"yseq": ys[i].tolist(), "score": float(sc[i]), "score_dec": float(sd[i]),
                 "score_ctc": float(scc[i]), "xpos": xp[i].tolist()} for i in range(n)]
