"""Server-side sessions on top of the batched scheduler: the per-client logic
of the reference's websocket server (SpeechRecognitionSession,
speechcatcher/speechcatcher_server.py:205-328) without its one-model-copy-per-
client pool - on-the-fly endpointing, Vosk-style JSON replies, eof / reset
control messages, the server's int16 -> float16/32767 input scaling (SURVEY
A10) - driven by ONE ``StreamScheduler`` so that the chunk steps of all
connected clients run as one batch on the GPU.

SURVEY.md section 8(f) rank 1.  The network layer (websockets, ffmpeg
transcoding of webm/mp3 input) stays out of scope: a transport hands
``ServerLoop.submit`` the messages it received and sends back what
``ServerLoop.step`` returns.

Differences to the reference, on purpose:
* token timestamps of final Vosk results are real (encoder frame of the token
  / 24 s, the CLI's convention speechcatcher.py:48) instead of the placeholder
  ``idx * 0.1`` (speechcatcher_server.py:309-311);
* after a finalised utterance, and for every new connection, the stream is reset
  (the native decoder of the reference keeps its finished state after
  ``is_final=True`` - the espnet decoder it replaced reset itself - so the reference
  server goes on decoding into a finalised stream, and a model returned to the pool
  hands its hypotheses to the next client).  ``ServerLoop(strict_reference=True)``
  switches this off and reproduces the reference server call for call;
* only 16 kHz s16le / int16 input (what Vosk clients send); other container
  formats need the transport to transcode first.
"""
import json
from collections import deque
from typing import Deque, Dict, List, Optional, Union

import numpy as np

from .scheduler import ServerBusy, StreamScheduler

Message = Union[str, bytes, np.ndarray]
FRAMES_PER_SECOND = 24.0   # speechcatcher.py:48


class Endpointer:
    """On-the-fly endpointing of one session (speechcatcher_server.py:252-268):
    an utterance is finalised when the best partial has had the same length for
    ``finalize_update_iters`` consecutive chunks, or after more than
    ``max_iters`` chunks; the history restarts after every finalisation."""

    def __init__(self, finalize_update_iters: int = 6, max_iters: int = 42):
        self.finalize_update_iters = finalize_update_iters
        self.max_iters = max_iters
        self.n_best_lens: List[int] = []

    def decide(self) -> bool:
        """Called BEFORE a chunk is decoded: finalise this chunk?"""
        n = len(self.n_best_lens)
        if n < self.finalize_update_iters:
            return False
        if n > self.max_iters:
            self.n_best_lens = []
            return True
        tail = self.n_best_lens[-self.finalize_update_iters:]
        if all(x == self.n_best_lens[-1] for x in tail):
            self.n_best_lens = []
            return True
        return False

    def observe(self, partial_len: int):
        """Called after a NON-final chunk that produced a result."""
        self.n_best_lens.append(partial_len)


def vosk_partial(text: str) -> dict:
    return {"partial": text}


def vosk_result(tokens: List[str], token_pos: Optional[List[int]] = None) -> dict:
    """Final result in Vosk style (speechcatcher_server.py:298-328): one entry
    per output token (not per word), "▁" is the sentencepiece space."""
    words, text = [], ""
    for idx, tok in enumerate(tokens):
        start = token_pos[idx] / FRAMES_PER_SECOND if token_pos is not None and idx < len(token_pos) else idx * 0.1
        words.append({"conf": 1.0, "start": start, "end": start + 1.0 / FRAMES_PER_SECOND,
                      "word": tok.replace("▁", " ")})
        text += tok
    return {"result": words, "text": text.replace("▁", " ").strip()}


def scale_server_pcm(data: np.ndarray) -> np.ndarray:
    """int16 -> the float values the reference server feeds its model:
    ``astype(float16) / 32767.0`` (rounded to float16), then fp32."""
    return (data.astype(np.float16) / np.float16(32767.0)).astype(np.float32)


class _Session:
    def __init__(self, sid: int, vosk: bool, finalize_update_iters: int, max_partial_iters: int):
        self.sid = sid
        self.vosk = vosk
        self.endpointer = Endpointer(finalize_update_iters, max_partial_iters)
        self.inbox: Deque[Message] = deque()
        self.in_flight: Optional[dict] = None     # the chunk currently queued in the scheduler
        self.vosk_sample_rate = 16000
        self.last = vosk_partial("") if vosk else ""


class ServerLoop:
    """Sessions of all connected clients over one ``StreamScheduler``."""

    def __init__(self, scheduler: StreamScheduler, vosk_output_format: bool = False,
                 finalize_update_iters: int = 6, max_partial_iters: int = 42, strict_reference: bool = False,
                 continuous: Optional[bool] = None, min_replies: int = 1):
        """``strict_reference``: no stream reset after a finalised utterance nor between clients, exactly like
        ``recognize_ws`` / ``process_audio_chunk`` (speechcatcher_server.py:270,359-397); the default resets.
        ``continuous`` (default: on whenever the batch has the C++ engine's submit / poll - round 4; False forces one
        batched lock-step call per step): a step hands the engine the chunks of the sessions that are ready and returns as soon as
        ``min_replies`` replies are (``StreamScheduler.pump``): every client is answered when ITS chunk is decoded
        and may send the next one at once, instead of all clients waiting for the slowest stream of a batch -
        the reference's per-client handler loop (:359-397), same calls and replies per session."""
        assert scheduler.result_format == "espnet", "sessions need token positions: result_format='espnet'"
        if strict_reference:
            scheduler.reset_after_final = scheduler.reset_on_open = False
        self.sch = scheduler
        if continuous is None:
            continuous = hasattr(scheduler.batch, "submit")
        self.continuous, self.min_replies = continuous, min_replies
        self.vosk = vosk_output_format
        self.fui, self.mpi = finalize_update_iters, max_partial_iters
        self.sessions: Dict[int, _Session] = {}

    # ---- connection lifecycle (recognize_ws, speechcatcher_server.py:359-397) ----
    def connect(self) -> int:
        """Raises ServerBusy when every stream slot is taken ("Server busy,
        please try again later.", :366)."""
        sid = self.sch.open()
        self.sessions[sid] = _Session(sid, self.vosk, self.fui, self.mpi)
        return sid

    def disconnect(self, sid: int):
        self.sessions.pop(sid)
        self.sch.close(sid)

    def submit(self, sid: int, message: Message):
        """Queue one message of a client (audio bytes / int16 array, or a control string)."""
        self.sessions[sid].inbox.append(message)

    # ---- one batched step -------------------------------------------------------
    def step(self) -> Dict[int, List[Union[str, dict]]]:
        """Feeds at most one audio chunk per session into the batch (the
        endpointing decision of chunk k needs the result of chunk k-1), runs one
        batched chunk step and returns the replies per session, in order.  An Exception instance
        among a session's replies means: that client's message could not be processed (its stream
        has been reset) - close that connection; no other session is affected."""
        replies: Dict[int, List[Union[str, dict]]] = {}
        for ses in self.sessions.values():
            while ses.inbox and ses.in_flight is None:
                try:
                    immediate = self._start(ses, ses.inbox.popleft())
                except (TypeError, NotImplementedError, ValueError) as exc:
                    replies.setdefault(ses.sid, []).append(exc)     # this client only
                    continue
                if immediate is not None:
                    replies.setdefault(ses.sid, []).append(self._reply(ses, immediate))
        for sid, results in (self.sch.pump(self.min_replies) if self.continuous else self.sch.step()).items():
            ses = self.sessions[sid]
            if isinstance(results, Exception):
                # the client's handler dies with the exception in the reference (recognize_ws has no except for
                # it): the transport closes this connection; every other session is untouched
                ses.in_flight = None
                ses.endpointer.n_best_lens = []
                replies.setdefault(sid, []).append(results)
                continue
            replies.setdefault(sid, []).append(self._reply(ses, self._finish(ses, results)))
        return replies

    def pending(self) -> bool:
        return any(s.inbox or s.in_flight for s in self.sessions.values())

    # ---- process_audio_chunk, split around the batched model call (:205-296) ----
    def _start(self, ses: _Session, message: Message):
        forced = False
        if isinstance(message, str):
            if not ses.vosk:
                return ""
            if message in ('{"eof" : 1}', '{"reset" : 1}'):
                forced = True
                data = np.zeros(1000, dtype=np.int16)
            else:
                try:
                    cfg = json.loads(message).get("config", {})
                    ses.vosk_sample_rate = int(cfg.get("sample_rate", ses.vosk_sample_rate))
                except (ValueError, AttributeError):
                    pass
                return vosk_partial("")
        elif isinstance(message, np.ndarray):
            if message.dtype != np.int16:
                raise TypeError("audio arrays must be int16 PCM")
            data = message
        else:
            if ses.vosk_sample_rate != 16000:
                raise NotImplementedError("transcode to 16 kHz s16le before submitting (ffmpeg path is out of scope)")
            data = np.frombuffer(message, dtype="<i2")
        if data.size == 0:
            return vosk_partial("") if ses.vosk else ""
        finalize = ses.endpointer.decide() or forced
        self.sch.feed(ses.sid, scale_server_pcm(data), is_final=finalize, finalize_all=False)
        ses.in_flight = {"finalize": finalize, "forced": forced}
        return None

    def _finish(self, ses: _Session, results: list):
        info, ses.in_flight = ses.in_flight, None
        if info["forced"]:
            ses.endpointer.n_best_lens = []   # session.reset() after a client-forced finalize (:272-273)
        if not results:
            return ""
        text, tokens, _ids, pos, _hyp = results[0]
        if info["finalize"]:
            if len(text) >= 1:
                if text[-1] not in ".!?":
                    text += "."
                text += "\n"
            return vosk_result(tokens, pos) if ses.vosk else text
        ses.endpointer.observe(len(text))
        return vosk_partial(text) if ses.vosk else text

    @staticmethod
    def _reply(ses: _Session, transcription):
        """recognize_ws (:376-393): Vosk clients get an answer for every message; an
        empty transcription repeats the last one, a final result only once."""
        if transcription:
            ses.last = transcription
            return transcription
        if ses.vosk:
            if isinstance(ses.last, dict) and "result" in ses.last:
                ses.last = vosk_partial("")
            return ses.last
        return transcription


class StepPacer:
    """WHEN to run the next batched step.  The reference answers every message with its own model call as it
    arrives (``recognize_ws``, speechcatcher_server.py:376-380); a batch needs a policy instead: run a step as
    soon as every connected session has a message waiting (a full batch), or when the oldest waiting message has
    waited ``max_wait_s`` (latency bound for the clients that did send), whichever comes first.  The transport
    calls ``submit`` for every received message and ``poll`` from its event loop."""

    def __init__(self, loop: ServerLoop, max_wait_s: float = 0.05, clock=None):
        import time
        self.loop = loop
        self.max_wait_s = max_wait_s
        self.clock = clock or time.monotonic
        self._stamps: Dict[int, Deque[float]] = {}

    def submit(self, sid: int, message: Message):
        self._stamps.setdefault(sid, deque()).append(self.clock())
        self.loop.submit(sid, message)

    def due(self) -> bool:
        sessions = self.loop.sessions
        waiting = [sid for sid, ses in sessions.items() if ses.inbox]
        if not waiting:
            return False
        if len(waiting) == len(sessions):
            return True
        oldest = min(self._stamps[sid][0] for sid in waiting if self._stamps.get(sid))
        return self.clock() - oldest >= self.max_wait_s

    def poll(self) -> Optional[Dict[int, List[Union[str, dict]]]]:
        """Runs one batched step if one is due and returns its replies (None otherwise)."""
        if not self.due():
            return None
        replies = self.loop.step()
        for sid in list(self._stamps):
            ses = self.loop.sessions.get(sid)
            if ses is None:                       # disconnected meanwhile
                del self._stamps[sid]
                continue
            st = self._stamps[sid]
            while len(st) > len(ses.inbox):       # messages the step consumed
                st.popleft()
        return replies


__all__ = ["Endpointer", "ServerLoop", "StepPacer", "ServerBusy", "vosk_partial", "vosk_result", "scale_server_pcm"]
