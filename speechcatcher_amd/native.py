"""The stream-level C ABI of libscasr.so (include/scasr.h: sc_engine_* / sc_streams_* / sc_push / sc_get_hyps /
sc_reset) behind the interface of ``engine.StreamBatch``: the whole state machine of the decoder - frontend and
encoder buffering, block schedule, beam-search step loop - runs in C++ (csrc/streams.hip); this file only
marshals arguments.  It is what ``Speech2TextStreaming`` / ``load_model`` / the scheduler use on a GPU.

``engine.StreamBatch`` is the same host logic in Python; it stays as the executable specification that runs on the
CPU spec backend against the reference fixtures (and as the home of the opt-in deferred-stragglers mode).
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
                 strict_reference: bool = True, engine: Optional[NativeEngine] = None, kv_dtype: str = "float32"):
        """``kv_dtype``: "float32" (the reference's arithmetic) or "float16" - the self- and cross-attention K|V
        caches in fp16 (half the HBM stream of the attention kernels and half the cache memory; arithmetic,
        softmax and all scores stay fp32): the storage mode of BASELINE configs[4], opt-in, never the parity mode."""
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
                               {"float32": 0, "float16": 1}[kv_dtype])
        self.kv_dtype = kv_dtype
        h = C.c_void_p()
        try:
            _abi.check(self.lib.sc_streams_create(self.engine.handle, C.byref(o), C.byref(h)), "sc_streams_create")
        except _abi.ScasrError as e:
            raise EngineError(str(e)) from e
        self.handle = h
        self.st = _InfoList(self)
        self.defer_threshold = 0

    # ---- StreamBatch interface ---------------------------------------------------------------------
    def set_defer_threshold(self, n_streams: int, max_lag_blocks: int = 1):
        if n_streams > 0:
            raise EngineError("deferred stragglers are a mode of the Python engine (engine.StreamBatch)")

    def flush(self):
        pass

    @property
    def stats(self):
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        self.lib.sc_streams_stats(self.handle, C.byref(a), C.byref(b), C.byref(c))
        d, e = C.c_long(), C.c_long()
        self.lib.sc_streams_speculation(self.handle, C.byref(d), C.byref(e))
        return {"enc_calls": a.value, "dec_steps": b.value, "dec_blocks": c.value,
                "spec_launched": d.value, "spec_wasted": e.value}

    def set_speculation(self, on: bool):
        """decode iterations enqueued ahead of their predecessor's stop flags (default on; results identical)"""
        _abi.check(self.lib.sc_streams_set_speculation(self.handle, 1 if on else 0), "sc_streams_set_speculation")

    def _call(self, fn, ids, ptrs, counts, finals, keep, isolate_faults):
        n = len(ids)
        a_ids = (C.c_int32 * n)(*ids)
        a_ptr = (C.c_void_p * n)(*ptrs)
        a_cnt = (C.c_int32 * n)(*counts)
        a_fin = (C.c_uint8 * n)(*[1 if f else 0 for f in finals])
        status = (C.c_int32 * n)()
        _abi.check(getattr(self.lib, fn)(self.handle, a_ids, a_ptr, a_cnt, a_fin, n, status), fn)
        del keep
        out = {}
        for i, s in enumerate(ids):
            if status[i] >= 0:
                out[s] = bool(status[i])
                continue
            msg = (self.lib.sc_last_error() or b"").decode()
            exc = (RuntimeError if status[i] == _ERR_INPUT else EngineError)(f"stream {s}: {msg}")
            if not isolate_faults:
                raise exc     # the other streams of the call were decoded; this one has been reset
            out[s] = exc
        return out

    def push(self, chunks: Sequence[Tuple[int, Optional[np.ndarray], bool]], pcm_resident: bool = False,
             prefetch=None, isolate_faults: bool = False):
        """One chunk step: (stream, samples, is_final); with ``pcm_resident`` the tuple carries the sample COUNT
        and the samples already sit in the device PCM ring (write_pcm).  Returns {stream: has_output}; a stream
        that failed carries its exception instead when ``isolate_faults`` (else it is raised)."""
        if prefetch is not None:
            raise EngineError("prefetch is a mode of the Python engine")
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
        return self._call("sc_push", ids, ptrs, counts, finals, keep, isolate_faults)

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

    def hypotheses(self, s: int):
        """Live hypotheses of stream s: list of dicts (yseq, score, score_dec, score_ctc, xpos), best first."""
        W, LC = self.W, self.LCAP
        ids = (C.c_int32 * (W * LC))()
        xp = (C.c_int32 * (W * LC))()
        lens = (C.c_int32 * W)()
        sc, sd, scc = (C.c_double * W)(), (C.c_double * W)(), (C.c_double * W)()
        n = self.lib.sc_get_hyps(self.handle, int(s), W, LC, ids, xp, lens, sc, sd, scc)
        if n < 0:
            _abi.check(n, "sc_get_hyps")
        return [{"yseq": list(ids[i * LC:i * LC + lens[i]]), "score": float(sc[i]), "score_dec": float(sd[i]),
                 "score_ctc": float(scc[i]), "xpos": list(xp[i * LC:i * LC + lens[i]])} for i in range(n)]

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

    def take_xattn_rows_by_kernel(self):
        """(rows read by the dec_attn_flash launches, rows read by the sc_dec_layer_cross launches)"""
        r = (C.c_long * 2)()
        _abi.check(self.lib.sc_streams_take_xattn_rows_by_kernel(self.handle, r), "sc_streams_take_xattn_rows_by_kernel")
        return int(r[0]), int(r[1])

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
