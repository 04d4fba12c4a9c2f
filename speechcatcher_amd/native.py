"""The stream-level C ABI of libscasr.so (include/scasr.h: sc_engine_* / sc_streams_* / sc_push / sc_submit / sc_poll /
sc_get_hyps_batch / sc_reset) behind the interface of ``engine.StreamBatch``: the whole state machine of the decoder - frontend and
encoder buffering, block schedule, beam-search step loop - runs in C++ (csrc/streams.hip); this file only
marshals arguments.  It is what ``Speech2TextStreaming`` / ``load_model`` / the scheduler use on a GPU.

``engine.StreamBatch`` is the same host logic in Python; it stays as the executable specification that runs on the
CPU spec backend against the reference fixtures.  Continuous batching (a stream's reply is ready when ITS blocks
are done, early streams start their next chunk while stragglers finish) is ``submit`` / ``poll`` here.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _abi
from .config import SearchConfig
from .engine import EngineError
from .weights import PackedWeights

_ERR_CAPACITY, _ERR_INPUT = -3, -4


class NativeEngine:
    """One weight replica on one GPU (sc_engine)."""

    def __init__(self, weights: Optional[PackedWeights] = None, packed_path: Optional[str] = None, device="cuda:0"):
        self.lib = _abi.load()
        self.device = torch.device(device)
        h = C.c_void_p()
        if packed_path is not None:
            _abi.check(self.lib.sc_engine_load(str(packed_path).encode(), self.device.index or 0, C.byref(h)),
                       "sc_engine_load")
            self.weights = None
        else:
            self.weights = weights            # keeps the device tensors alive: the engine borrows them
            cfg = weights.cfg
            c = _abi.Config()
            for n, _ in _abi.Config._fields_:
                setattr(c, n, weights.mvn_mode() if n == "mvn_mode" else getattr(cfg, n))
            ts = weights.named_tensors()
            self._names = [n.encode() for n, _ in ts]
            arr = (_abi.NamedTensor * len(ts))()
            for i, (n, t) in enumerate(ts):
                arr[i].name, arr[i].data, arr[i].numel = self._names[i], t.data_ptr(), t.numel()
                arr[i].dtype = {torch.float64: 1, torch.float16: 2}.get(t.dtype, 0)
            _abi.check(self.lib.sc_engine_create(C.byref(c), arr, len(ts), self.device.index or 0, C.byref(h)),
                       "sc_engine_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sc_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass


class _Info:
    __slots__ = ("_b", "_s")
    _MAP = {"T_enc": "enc_frames", "processed_block": "processed_block", "process_idx": "process_idx",
            "L": "hyp_len", "nhyp": "n_hyp", "n_steps_total": "decode_steps", "fe_started": "frontend_started",
            "pcm_buffered": "pcm_buffered"}

    def __init__(self, b, s):
        self._b, self._s = b, s

    def __getattr__(self, name):
        info = _abi.StreamInfo()
        _abi.check(self._b.lib.sc_stream_info(self._b.handle, self._s, C.byref(info)), "sc_stream_info")
        v = getattr(info, self._MAP[name])
        return bool(v) if name == "fe_started" else int(v)


class _InfoList:
    def __init__(self, b):
        self._b = b

    def __getitem__(self, s):
        return _Info(self._b, s)

    def __iter__(self):
        return (_Info(self._b, s) for s in range(self._b.S))

    def __len__(self):
        return self._b.S


class NativeStreamBatch:
    """S independent streams on one engine (sc_streams); interface of engine.StreamBatch."""

    def __init__(self, weights, n_streams: int, search: SearchConfig = SearchConfig(), max_frames: int = 1600,
                 max_tokens: int = 640, pcm_capacity: int = 1 << 20, max_chunk_samples: int = 32768,
                 strict_reference: bool = True, engine: Optional[NativeEngine] = None, kv_dtype: str = "float32",
                 kv_pool_rows: int = 0):
        """``kv_dtype``: "float32" (the reference's arithmetic) or "float16" - the self- and cross-attention K|V
        caches in fp16 (half the HBM stream of the attention kernels and half the cache memory; arithmetic,
        softmax and all scores stay fp32): the storage mode of BASELINE configs[4], opt-in, never the parity mode.
        ``kv_pool_rows``: rows of the self-attention K|V pool per stream and layer (0: one row per (position, hypothesis) -
        never exhausted before max_tokens - unless that takes more than a quarter of the free device memory, then what the
        budget holds, at least 1.5 x max_tokens + 4 x beam; a stream that needs more fails with a capacity error)."""
        if not torch.cuda.is_available():
            raise _abi.ScasrError("NativeStreamBatch needs a ROCm GPU (torch.cuda.is_available() is False)")
        self.engine = engine or NativeEngine(weights, device=weights.device)
        self.lib = self.engine.lib
        self.w, self.cfg, self.search = weights, (weights.cfg if weights is not None else None), search
        self.S, self.W = n_streams, search.beam_size
        self.TCAP, self.LCAP, self.PCAP = max_frames, max_tokens, pcm_capacity
        self.strict_reference = strict_reference
        if search.pre_beam != 40 or search.max_length != 500:
            raise EngineError("the native engine implements the reference's constants: pre-beam 40, max_length 500")
        o = _abi.StreamOptions(n_streams, search.beam_size, search.ctc_weight, int(search.use_bbd), max_frames,
                               max_tokens, pcm_capacity, max_chunk_samples, int(strict_reference),
                               {"float32": 0, "float16": 1}[kv_dtype], int(kv_pool_rows))
        self.kv_dtype = kv_dtype
        h = C.c_void_p()
        try:
            _abi.check(self.lib.sc_streams_create(self.engine.handle, C.byref(o), C.byref(h)), "sc_streams_create")
        except _abi.ScasrError as e:
            raise EngineError(str(e)) from e
        self.handle = h
        self.st = _InfoList(self)

    # ---- StreamBatch interface ---------------------------------------------------------------------
    @property
    def stats(self):
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        self.lib.sc_streams_stats(self.handle, C.byref(a), C.byref(b), C.byref(c))
        return {"enc_calls": a.value, "dec_steps": b.value, "dec_blocks": c.value}

    def _fault(self, s, code):
        msg = (self.lib.sc_stream_last_error(self.handle, int(s)) or b"").decode()
        return (RuntimeError if code == _ERR_INPUT else EngineError)(f"stream {s}: {msg}")

    @staticmethod
    def _marshal(ids, ptrs, counts, finals):
        n = len(ids)
        return ((C.c_int32 * n)(*ids), (C.c_void_p * n)(*ptrs), (C.c_int32 * n)(*counts),
                (C.c_uint8 * n)(*[1 if f else 0 for f in finals]))

    def _call(self, fn, ids, ptrs, counts, finals, keep, isolate_faults):
        n = len(ids)
        a_ids, a_ptr, a_cnt, a_fin = self._marshal(ids, ptrs, counts, finals)
        status = (C.c_int32 * n)()
        try:
            _abi.check(getattr(self.lib, fn)(self.handle, a_ids, a_ptr, a_cnt, a_fin, n, status), fn)
        except _abi.ScasrError as e:
            raise EngineError(str(e)) from e
        del keep
        out = {}
        for i, s in enumerate(ids):
            if status[i] >= 0:
                out[s] = bool(status[i])
                continue
            exc = self._fault(s, status[i])
            if not isolate_faults:
                raise exc     # the other streams of the call were decoded; this one has been reset
            out[s] = exc
        return out

    def _gather(self, chunks, pcm_resident):
        ids, ptrs, counts, finals, keep = [], [], [], [], []
        for s, samples, fin in chunks:
            ids.append(int(s))
            finals.append(bool(fin))
            if pcm_resident:
                ptrs.append(None)
                counts.append(int(samples))
            else:
                if isinstance(samples, torch.Tensor):
                    samples = samples.detach().cpu().numpy()
                a = np.ascontiguousarray(samples, dtype=np.float32)
                keep.append(a)
                ptrs.append(a.ctypes.data)
                counts.append(int(a.shape[0]))
        return ids, ptrs, counts, finals, keep

    def push(self, chunks: Sequence[Tuple[int, Optional[np.ndarray], bool]], pcm_resident: bool = False,
             isolate_faults: bool = False):
        """One chunk step: (stream, samples, is_final); with ``pcm_resident`` the tuple carries the sample COUNT
        and the samples already sit in the device PCM ring (write_pcm).  Returns {stream: has_output}; a stream
        that failed carries its exception instead when ``isolate_faults`` (else it is raised)."""
        ids, ptrs, counts, finals, keep = self._gather(chunks, pcm_resident)
        return self._call("sc_push", ids, ptrs, counts, finals, keep, isolate_faults)

    def push_block(self, stream_ids: np.ndarray, pcm: np.ndarray, finals: Optional[np.ndarray] = None) -> np.ndarray:
        """sc_push for row i of a contiguous float32 matrix ``pcm`` [n, samples] -> stream ``stream_ids[i]`` without
        per-chunk Python work (the batched callers: bench, scheduler).  Returns the int32 status array (sc_push)."""
        n, m = pcm.shape
        assert pcm.dtype == np.float32 and pcm.flags.c_contiguous and len(stream_ids) == n
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        ptrs = (pcm.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(m * 4)).astype(np.uint64)
        cnt = np.full(n, m, np.int32)
        fin = np.zeros(n, np.uint8) if finals is None else np.ascontiguousarray(finals, dtype=np.uint8)
        status = np.zeros(n, np.int32)
        _abi.check(self.lib.sc_push(self.handle, ids.ctypes.data_as(_abi.c_int_p), ptrs.ctypes.data_as(C.POINTER(C.c_void_p)),
                                    cnt.ctypes.data_as(_abi.c_int_p), fin.ctypes.data_as(C.POINTER(C.c_uint8)), n,
                                    status.ctypes.data_as(_abi.c_int_p)), "sc_push")
        return status

    # ---- continuous batching (sc_submit / sc_poll) ---------------------------------------------------
    def submit(self, chunks: Sequence[Tuple[int, Optional[np.ndarray], bool]], pcm_resident: bool = False):
        """Hand the engine one chunk per listed stream and return at once (the chunks are copied); their frontend +
        encoder stage is issued as one group.  A stream's next chunk may be submitted once ``poll`` has reported
        this one."""
        ids, ptrs, counts, finals, keep = self._gather(chunks, pcm_resident)
        a_ids, a_ptr, a_cnt, a_fin = self._marshal(ids, ptrs, counts, finals)
        try:
            _abi.check(self.lib.sc_submit(self.handle, a_ids, a_ptr, a_cnt, a_fin, len(ids)), "sc_submit")
        except _abi.ScasrError as e:
            raise EngineError(str(e)) from e
        del keep

    def submit_block(self, stream_ids: np.ndarray, pcm: np.ndarray, finals: Optional[np.ndarray] = None):
        n, m = pcm.shape
        assert pcm.dtype == np.float32 and pcm.flags.c_contiguous and len(stream_ids) == n
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        ptrs = (pcm.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(m * 4)).astype(np.uint64)
        cnt = np.full(n, m, np.int32)
        fin = np.zeros(n, np.uint8) if finals is None else np.ascontiguousarray(finals, dtype=np.uint8)
        _abi.check(self.lib.sc_submit(self.handle, ids.ctypes.data_as(_abi.c_int_p), ptrs.ctypes.data_as(C.POINTER(C.c_void_p)),
                                      cnt.ctypes.data_as(_abi.c_int_p), fin.ctypes.data_as(C.POINTER(C.c_uint8)), n), "sc_submit")

    def poll(self, min_done: int = 1, isolate_faults: bool = True):
        """Run the engine until at least ``min_done`` outstanding chunks are complete (or none is outstanding).
        Returns {stream: has_output | exception} for the chunks reported by this call."""
        ids = np.zeros(self.S, np.int32)
        st = np.zeros(self.S, np.int32)
        n = self.lib.sc_poll(self.handle, int(min_done), self.S, ids.ctypes.data_as(_abi.c_int_p), st.ctypes.data_as(_abi.c_int_p))
        if n < 0:
            _abi.check(n, "sc_poll")
        out = {}
        for i in range(n):
            s = int(ids[i])
            if st[i] >= 0:
                out[s] = bool(st[i])
            else:
                exc = self._fault(s, int(st[i]))
                if not isolate_faults:
                    raise exc
                out[s] = exc
        return out

    def poll_ids(self, min_done: int = 1, max_done: int = 0):
        """``poll`` without Python objects: (stream ids, statuses) as int32 arrays; at most ``max_done`` replies
        (0: all that are ready), oldest completion first."""
        ids = np.zeros(self.S, np.int32)
        st = np.zeros(self.S, np.int32)
        n = self.lib.sc_poll(self.handle, int(min_done), int(max_done) if max_done > 0 else self.S,
                             ids.ctypes.data_as(_abi.c_int_p), st.ctypes.data_as(_abi.c_int_p))
        if n < 0:
            _abi.check(n, "sc_poll")
        return ids[:n], st[:n]

    def set_encoder_batch(self, min_streams: int):
        """continuous batching: the encoder stages of successive admissions are issued as one group when it holds this
        many streams (or as soon as a decode block needs its frames); results do not depend on it"""
        _abi.check(self.lib.sc_streams_set_encoder_batch(self.handle, int(min_streams)), "sc_streams_set_encoder_batch")

    def set_queue_depth(self, depth: int):
        """chunks a stream may have outstanding (default 1: call -> reply -> next call).  depth > 1: ``submit`` accepts the
        next chunk(s) of a stream while an earlier one is still decoding (file / backlog hosts: the encoder of chunk k+1
        runs beside the decoding of chunk k); chunks are reported in order, one per ``poll`` and stream, with the same
        results, and the hypotheses of a reported chunk stay readable until the next ``poll``"""
        _abi.check(self.lib.sc_streams_set_queue_depth(self.handle, int(depth)), "sc_streams_set_queue_depth")
        self.queue_depth = int(depth)

    @property
    def outstanding(self) -> int:
        return int(self.lib.sc_streams_outstanding(self.handle))

    def push_features(self, items, isolate_faults: bool = False):
        ids, ptrs, counts, finals, keep = [], [], [], [], []
        for s, feats, fin in items:
            if isinstance(feats, torch.Tensor):
                feats = feats.detach().cpu().numpy()
            a = np.ascontiguousarray(feats, dtype=np.float32)
            if a.ndim != 2 or a.shape[1] != self.cfg.n_mels:
                raise EngineError(f"features must be (T, {self.cfg.n_mels})")
            keep.append(a)
            ids.append(int(s)); ptrs.append(a.ctypes.data); counts.append(int(a.shape[0])); finals.append(bool(fin))
        return self._call("sc_push_features", ids, ptrs, counts, finals, keep, isolate_faults)

    def reset(self, s: int):
        _abi.check(self.lib.sc_reset(self.handle, int(s)), "sc_reset")

    def reset_all(self):
        for s in range(self.S):
            self.reset(s)

    def hypotheses_arrays(self, streams: Sequence[int], nbest: Optional[int] = None):
        """Live hypotheses of the listed streams with ONE device round trip (sc_get_hyps_batch): numpy arrays
        ids / xpos [n, nbest, Lmax], lens / scores / score_dec / score_ctc [n, nbest], n_hyps [n]."""
        nb = self.W if nbest is None else int(nbest)
        sid = np.ascontiguousarray(streams, dtype=np.int32)
        n = len(sid)
        # (queue depth > 1: the copies taken at completion may be longer than the stream's live hypotheses)
        lmax = max([1] + [self.st[int(s)].L for s in sid]) if n <= 4 and getattr(self, "queue_depth", 1) <= 1 else self.LCAP
        ids = np.zeros((n, nb, lmax), np.int32)
        xp = np.zeros((n, nb, lmax), np.int32)
        lens = np.zeros((n, nb), np.int32)
        nh = np.zeros(n, np.int32)
        sc, sd, scc = np.zeros((n, nb)), np.zeros((n, nb)), np.zeros((n, nb))
        ip, dp = _abi.c_int_p, _abi.c_double_p
        _abi.check(self.lib.sc_get_hyps_batch(self.handle, sid.ctypes.data_as(ip), n, nb, lmax, ids.ctypes.data_as(ip),
                                              xp.ctypes.data_as(ip), lens.ctypes.data_as(ip), nh.ctypes.data_as(ip),
                                              sc.ctypes.data_as(dp), sd.ctypes.data_as(dp), scc.ctypes.data_as(dp)),
                   "sc_get_hyps_batch")
        return {"ids": ids, "xpos": xp, "lens": lens, "n_hyps": nh, "score": sc, "score_dec": sd, "score_ctc": scc}

    def hypotheses_batch(self, streams: Sequence[int], nbest: Optional[int] = None):
        """{stream: [hypothesis dicts, best first]} for the listed streams, one device round trip for all."""
        a = self.hypotheses_arrays(streams, nbest)
        out = {}
        for i, s in enumerate(streams):
            hy = []
            for j in range(int(a["n_hyps"][i])):
                L = int(a["lens"][i, j])
                hy.append({"yseq": a["ids"][i, j, :L].tolist(), "score": float(a["score"][i, j]),
                           "score_dec": float(a["score_dec"][i, j]), "score_ctc": float(a["score_ctc"][i, j]),
                           "xpos": a["xpos"][i, j, :L].tolist()})
            out[int(s)] = hy
        return out

    def hypotheses(self, s: int):
        """Live hypotheses of stream s: list of dicts (yseq, score, score_dec, score_ctc, xpos), best first."""
        return self.hypotheses_batch([int(s)])[int(s)]

    # ---- device-resident audio (bench) and the views of the drop-in class -------------------------------
    def write_pcm(self, s: int, offset: int, samples: np.ndarray):
        a = np.ascontiguousarray(samples, dtype=np.float32)
        _abi.check(self.lib.sc_streams_write_pcm(self.handle, int(s), int(offset), a.ctypes.data, a.shape[0]),
                   "sc_streams_write_pcm")

    def waveform_buffer(self, s: int) -> np.ndarray:
        n = self.st[s].pcm_buffered
        a = np.zeros(n, np.float32)
        if n:
            self.lib.sc_streams_read_pcm_buffer(self.handle, int(s), a.ctypes.data, n)
        return a

    def encoder_buffer(self, s: int) -> Optional[np.ndarray]:
        T = self.st[s].T_enc
        if T == 0:
            return None
        a = np.zeros((T, self.cfg.d_model), np.float32)
        self.lib.sc_streams_read_enc(self.handle, int(s), a.ctypes.data, T)
        return a

    def set_graphs(self, on: bool):
        self.lib.sc_streams_set_graphs(self.handle, int(bool(on)))

    def take_xattn_rows(self) -> int:
        return int(self.lib.sc_streams_take_xattn_rows(self.handle))

    def take_attn_counters(self):
        """{cross_rows, self_positions, self_distinct_rows}: [stand-alone kernels, head-parallel layer kernels, stream-resident
        layer kernel], summed over decode iterations, active streams and decoder layers since the last call"""
        r = (C.c_long * 9)()
        _abi.check(self.lib.sc_streams_take_attn_counters(self.handle, r), "sc_streams_take_attn_counters")
        return {"cross_rows": [int(r[0]), int(r[1]), int(r[2])], "self_positions": [int(r[3]), int(r[4]), int(r[5])],
                "self_distinct_rows": [int(r[6]), int(r[7]), int(r[8])]}

    def take_xattn_rows_by_kernel(self):
        """(rows read by the dec_attn_flash launches, by the sc_dec_layer_cross launches, by the sc_dec_layer_stream launches)"""
        r = (C.c_long * 3)()
        _abi.check(self.lib.sc_streams_take_xattn_rows_by_kernel(self.handle, r), "sc_streams_take_xattn_rows_by_kernel")
        return int(r[0]), int(r[1]), int(r[2])

    @property
    def kv_rows(self) -> int:
        """rows of the self-attention K|V pool per (stream, layer) this batch was created with"""
        return int(self.lib.sc_streams_kv_rows(self.handle))

    @property
    def hip_stream(self) -> int:
        return int(self.lib.sc_streams_hip_stream(self.handle) or 0)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sc_streams_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
