#!/usr/bin/env python3
"""Benchmark of the streaming-ASR hot path on MI355X.

One "step" = one chunk step (10 240 samples = 640 ms of 16 kHz audio) of the whole path - log-mel frontend,
Conv2d subsampling, contextual-block encoder, blockwise-synchronous beam search (decoder + CTC prefix scorer) - for
EVERY stream of the batch (S streams per GPU, default 128 = BASELINE.json configs[2]; `--streams 1` gives
configs[1]), run to completion inside the step: every stream's result is available when its call returns, exactly
like the reference's per-call semantics (STRICT lock-step).  The step runs on the C++ engine behind the stream-level
C ABI (sc_push + sc_get_hyps_batch; csrc/streams.hip) - the product path of Speech2TextStreaming and of the scheduler.

The timed region contains the BOUNDARY the reference's step has (speech2text_streaming.py:402-539): every step takes
its 128 chunks from HOST memory (pinned staging, one H2D copy) and hands the best hypothesis of every stream (token
ids + encoder-frame positions + scores) back to the host (one pack launch + one D2H copy).  The same window with the
audio resident in HBM and no read-back is reported under `resident_no_readback`.

Window: the streams are first rolled (untimed) to the MIDDLE of SURVEY 8(d)'s 60 s utterance - `--preroll` 21 steps
in front of the warm-up, so that with the driver's `--warmup 5 --steps 20` the timed steps are hops 27-46 of every
stream (T = 430..740 encoder frames, 200..350 tokens per hypothesis) instead of its cheap beginning.

Weights: de_streaming_transformer_xl dimensions, seeded synthetic (no checkpoints offline); audio: seeded Gaussian
noise.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...
    python bench.py --gpus 8 ...        (no launcher: bench.py starts the 8 rank processes itself, rank i on GPU i)

Prints ONE JSON line (rank 0).  value = whole-job audio-seconds processed per wall-clock second = number of
concurrent real-time streams the job sustains.  Extra keys: `served` (continuous batching through sc_submit / sc_poll:
the same chunks, the same per-call results, every stream gets its next chunk as soon as its previous reply has been
delivered), `resident_no_readback`, `long_context` (T ~ 1000 and T ~ 4500 encoder frames), `roofline`,
`cpu_baseline`, `single_stream`, `whole_step`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from speechcatcher_amd import synth  # noqa: E402
from speechcatcher_amd.config import XL, SearchConfig  # noqa: E402

CHUNK = 10240
SERVED_SPARE = 8               # chunks per stream beyond the window (served leg: streams that run ahead of the average)
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # same guide: HBM3E ~8 TB/s
# algorithmic work of one stream-hop (10 240 samples of one stream)
GFLOP_ENCODER_SIDE_PER_HOP = 3.83
GFLOP_PER_DECODE_STEP = 0.437  # at beam 10


def capacities(n_steps_total):
    """encoder frames / tokens per hypothesis a window of n steps can reach (the reference's step loop is bounded by
    process_idx < 500, beam_search.py:701: from then on a hypothesis grows by ONE token per block - the step that
    reaches the bound is kept, the rewind only takes process_idx back)"""
    hops = (n_steps_total + 2) * CHUNK / 10240.0       # encoder hops (16 frames each) in the window
    return int(16 * hops) + 64, min(int(14 * hops) + 32, 540 + int(hops))


FFN_DTYPE = "float32"  # --ffn-dtype float16: fp16 feed-forward weights + fp16 MFMA inputs (configs[4]), never the default
ENCODER_BATCH = 0      # --encoder-batch N: sc_streams_set_encoder_batch (0: the engine's default, half of the streams)
KV_DTYPE = "float32"   # --kv-dtype float16: K|V caches in fp16 (BASELINE configs[4]'s storage mode), never the default


def make_weights(device, ffn_dtype=None, cfg=XL):
    from speechcatcher_amd.weights import PackedWeights
    ffn_dtype = ffn_dtype or FFN_DTYPE
    sd = synth.make_state_dict(cfg, 1234)
    mean, std = synth.stats_to_mean_std(synth.make_stats(cfg, kind="meanstd"))
    return PackedWeights(sd, cfg, device, mean, std, ffn_dtype=ffn_dtype,
                         proj_dtype=ffn_dtype,   # the encoder's attention projections in the same form
                         # float16 = the whole fp16 mode of BASELINE configs[4]: the decoder's projections / output layer and
                         # the partial products between its kernels as well (effective together with --kv-dtype float16)
                         dec_dtype="float16" if ffn_dtype == "float16" else "float32")


def build_native(w, n_streams, beam, bbd, n_steps_total, engine=None, kv_dtype=None):
    from speechcatcher_amd.native import NativeStreamBatch
    frames, tokens = capacities(n_steps_total)
    sb = NativeStreamBatch(w, n_streams, SearchConfig(beam_size=beam, use_bbd=bbd), max_frames=frames,
                           max_tokens=tokens, pcm_capacity=CHUNK * (n_steps_total + 2), max_chunk_samples=CHUNK,
                           engine=engine, kv_dtype=kv_dtype or KV_DTYPE)
    if ENCODER_BATCH > 0:
        sb.set_encoder_batch(min(ENCODER_BATCH, n_streams))
    return sb


def make_audio(n_streams, n_steps, stream_offset=0, shared=False):
    """[n_streams][n_steps * CHUNK] float32.  Per-stream seeded noise; shared=True: windows of ONE long
    seeded noise buffer at per-stream offsets (long-context runs: 128 x 180 s of torch.randn would take minutes)."""
    n = CHUNK * n_steps
    if not shared:
        return np.stack([synth.synth_audio(stream_offset + s, n) for s in range(n_streams)])
    base = synth.synth_audio(stream_offset, n + 7919 * n_streams)
    return np.stack([base[7919 * s:7919 * s + n] for s in range(n_streams)])


def preload_audio(sb, audio, upto_step):
    """the first `upto_step` chunks of every stream -> the device PCM ring (for the untimed pre-roll / resident mode)"""
    for s in range(sb.S):
        sb.write_pcm(s, 0, audio[s, :upto_step * CHUNK])
    torch.cuda.synchronize()


def run_resident(sb, n):
    items = [(s, CHUNK, False) for s in range(sb.S)]
    for _ in range(n):
        sb.push(items, pcm_resident=True)


def step_blocks(audio, k0, k1):
    """contiguous [S][CHUNK] host blocks of steps k0..k1-1 (what a transport hands the engine every 640 ms)"""
    return [np.ascontiguousarray(audio[:, k * CHUNK:(k + 1) * CHUNK]) for k in range(k0, k1)]


def run_host(sb, blocks, ids):
    """the served chunk step: host PCM in (pinned staging + one H2D inside sc_push), best hypothesis of every stream
    out (ids + positions + scores; one pack launch + one D2H inside sc_get_hyps_batch)"""
    out = None
    for blk in blocks:
        st = sb.push_block(ids, blk)
        assert (st >= 0).all()
        out = sb.hypotheses_arrays(ids, nbest=1)
    return out


def n_dec_steps(sb):
    return sum(st.n_steps_total for st in sb.st)


def roll(sb, audio, preroll):
    """the untimed pre-roll: `preroll` lock-step chunk steps on resident audio"""
    preload_audio(sb, audio, preroll)
    run_resident(sb, preroll)
    torch.cuda.synchronize()


def strict_window(sb, audio, k0, steps, dist=None, before_timing=None):
    """STRICT lock-step: `steps` batched sc_push calls (chunks k0.. of every stream), every block completes inside its
    call; host PCM in, best hypothesis of every stream out after every call."""
    ids = np.arange(sb.S, dtype=np.int32)
    blocks = step_blocks(audio, k0, k0 + steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    steps0, iters0, enc0 = n_dec_steps(sb), sb.stats["dec_steps"], sb.stats["enc_calls"]
    sb.take_attn_counters()
    if before_timing:
        before_timing(sb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run_host(sb, blocks, ids)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    extra = {"attn": sb.take_attn_counters(), "encoder_groups": sb.stats["enc_calls"] - enc0,
             "iterations": sb.stats["dec_steps"] - iters0}
    return elapsed, (n_dec_steps(sb) - steps0) / float(sb.S) / max(steps, 1), last, extra


def serve(sb, a3, nxt, steps, group, dist=None, boundary=True, before_timing=None, at_target=None, depth=1, exact=False):
    """CONTINUOUS batching, closed loop: every stream is served on its own - sc_submit its chunk, sc_poll until replies
    are ready, read the best hypothesis of the streams that answered (sc_get_hyps_batch), submit THEIR next chunks
    (stream i continues at chunk nxt[i] of a3 [S][chunks][CHUNK]).  boundary: host PCM in (pinned staging, one H2D
    per admission) and hypotheses out per reply; False: chunks already resident in the device PCM ring, no read-back.
    The clock stops when S x steps replies have been delivered - `steps` chunk steps' worth of audio; a stream whose
    blocks need fewer decode steps gets further than one that needs more (a3 must hold spare chunks).  What is in
    flight then is drained untimed.  depth > 1 (sc_streams_set_queue_depth): every stream keeps `depth` chunks with the
    engine - its next chunk is submitted while the previous one is still decoding (a host that has the audio already).
    exact: EVERY stream gets exactly `steps` chunks (a stream that has had its share is not fed again): the same S x steps
    replies, every stream with the same weight - the tail runs at thinning buckets like any real end of a batch of calls."""
    S = sb.S
    end = a3.shape[1]
    k_start = nxt.copy()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    steps0, iters0, enc0 = n_dec_steps(sb), sb.stats["dec_steps"], sb.stats["enc_calls"]
    sb.take_attn_counters()
    if before_timing:
        before_timing(sb)

    def submit(streams):
        streams = streams[nxt[streams] < end]
        if exact:
            streams = streams[nxt[streams] - k_start[streams] < steps]
        if len(streams):
            if boundary:
                sb.submit_block(streams, a3[streams, nxt[streams]])
            else:
                sb.submit([(int(i), CHUNK, False) for i in streams], pcm_resident=True)
            nxt[streams] += 1

    t0 = time.perf_counter()
    for _ in range(depth):
        submit(np.arange(S, dtype=np.int32))
    n_polls, n_replies, target, res = 0, 0, S * steps, None
    while sb.outstanding:
        done, st = sb.poll_ids(min(group, sb.outstanding))
        assert (st >= 0).all()
        n_polls += 1
        if boundary:
            sb.hypotheses_arrays(done, nbest=1)
        n_replies += len(done)
        if n_replies >= target:
            if res is None:
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t0
                attn_cnt = sb.take_attn_counters()
                if at_target:
                    at_target(attn_cnt)
                adv = nxt - k_start
                res = {"elapsed": elapsed, "dec_steps_per_hop": (n_dec_steps(sb) - steps0) / float(n_replies),
                       "attn": attn_cnt, "encoder_groups": sb.stats["enc_calls"] - enc0,
                       "iterations": sb.stats["dec_steps"] - iters0,
                       "iterations_per_step": (sb.stats["dec_steps"] - iters0) / float(steps), "polls_per_step": n_polls / float(steps),
                       "chunks_per_stream_min_max": [int(adv.min()), int(adv.max())]}
            continue          # (drain what is in flight, untimed)
        submit(done)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    assert res is not None, "the streams ran out of audio before the target number of replies"
    return res


def state_of(sb):
    a = sb.hypotheses_arrays(list(range(sb.S)), nbest=1)
    T = [st.T_enc for st in sb.st]
    return {"encoder_frames_T": [min(T), max(T)], "tokens_L": [int(a["lens"].min()), int(a["lens"].max())],
            "process_idx": [min(st.process_idx for st in sb.st), max(st.process_idx for st in sb.st)]}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (restatement of the reference path, validated against the reference's fixtures) timed on
# this box's host cores.  Three legs: one process with 1 and with 8 intra-op threads (the best
# single-process setting measured for the reference), and N independent single-thread
# processes - the reference's own concurrency model (one process / model copy per worker or client:
# speechcatcher.py:482,787, speechcatcher_server.py:331-357).
class _StatePickler(__import__("pickle").Pickler):
    """an oracle stream WITHOUT its model: the model object, its weight dict and its PE table go out as tags"""

    def __init__(self, f, model):
        super().__init__(f, protocol=4)
        self._tags = {id(model): "model", id(model.sd): "sd", id(model.pe): "pe"}

    def persistent_id(self, obj):
        return self._tags.get(id(obj))


class _StateUnpickler(__import__("pickle").Unpickler):
    def __init__(self, f, model):
        super().__init__(f)
        self._objs = {"model": model, "sd": model.sd, "pe": model.pe}

    def persistent_load(self, tag):
        return self._objs[tag]


def _oracle_model():
    from oracle.ref_port import RefPortModel
    from speechcatcher_amd.mel import melscale_fbanks_slaney
    sd = synth.make_state_dict(XL, 1234)
    mean, std = synth.stats_to_mean_std(synth.make_stats(XL, kind="meanstd"))
    return RefPortModel(sd, XL, melscale_fbanks_slaney(257, 0.0, 8000.0, 80, 16000), mean, std)


def cpu_fast_forward_worker(args):
    """untimed: stream 0 through the chunks IN FRONT of the GPU's timed window, state -> file"""
    threads, beam, bbd, k0, path = args
    import torch as th
    th.set_num_threads(threads)
    from oracle.ref_port import RefPortStreaming
    model = _oracle_model()
    s = RefPortStreaming(model, beam_size=beam, ctc_weight=0.3, use_bbd=bbd)
    audio = synth.synth_audio(0, CHUNK * k0)
    t0 = time.perf_counter()
    for i in range(k0):
        s(audio[i * CHUNK:(i + 1) * CHUNK], is_final=False)
    with open(path, "wb") as f:
        _StatePickler(f, model).dump(s)
    T = 0 if s.encoder_buffer is None else int(s.encoder_buffer.shape[1])
    return time.perf_counter() - t0, T, max(len(h.yseq) for h in s.running_hyps)


def cpu_baseline_worker(args):
    threads, budget_s, k0, max_steps, path = args
    import torch as th
    th.set_num_threads(threads)
    model = _oracle_model()
    with open(path, "rb") as f:
        s = _StateUnpickler(f, model).load()
    audio = synth.synth_audio(0, CHUNK * (k0 + max_steps))
    t0 = time.perf_counter()
    n = 0
    while n < max_steps and time.perf_counter() - t0 < budget_s:
        i = k0 + n
        s(audio[i * CHUNK:(i + 1) * CHUNK], is_final=False)
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(k0, budget_s=10.0, beam=10, bbd=False, max_steps=20):
    """The oracle on the SAME window as the GPU's timed region: chunks k0.. of stream 0 (T and the hypothesis lengths
    are what they are there - attention and the CTC scan walk the whole prefix).  The prefix is fast-forwarded once,
    untimed, and the state handed to every timed process."""
    import multiprocessing as mp
    import tempfile
    ncpu = os.cpu_count() or 1
    hop_s = CHUNK / 16000.0
    legs = []
    ctx = mp.get_context("spawn")       # fresh interpreters: never fork a process that has initialised the GPU
    fd, path = tempfile.mkstemp(suffix=".oracle_state")
    os.close(fd)
    try:
        with ctx.Pool(1) as pool:
            ff_s, T0, L0 = pool.map(cpu_fast_forward_worker, [(min(8, ncpu), beam, bbd, k0, path)])[0]
        for threads, procs in ((8, 1), (1, 1), (1, min(16, ncpu))):
            threads = min(threads, ncpu)
            with ctx.Pool(procs) as pool:
                # (the quoted leg gets the whole budget, the two side legs 60 % of it: the default command stays under 90 s)
                res = pool.map(cpu_baseline_worker, [(threads, budget_s if not legs else 0.6 * budget_s, k0, max_steps, path) for _ in range(procs)])
            rate = sum(n * hop_s / dt for n, dt in res)
            legs.append({"processes": procs, "threads_per_process": threads, "audio_s_per_s": round(rate, 4),
                         "steps": [n for n, _ in res][:4], "wall_s": round(max(dt for _, dt in res), 1)})
    finally:
        os.unlink(path)
    best = legs[0]
    return {"value": best["audio_s_per_s"], "unit": "audio_s/s", "cores": best["threads_per_process"], "kind": "port",
            "sample": f"1 stream, chunk steps {k0}.. of stream 0 - the GPU's timed window (T = {T0} encoder frames, "
                      f"{L0} tokens per hypothesis at its start; the {k0} chunks in front of it fast-forwarded untimed in "
                      f"{ff_s:.0f} s) - {best['steps'][0]} steps, {best['wall_s']} s wall, "
                      f"beam {beam}, bbd {int(bbd)}, XL dims, torch-CPU oracle (oracle/ref_port.py), 8 intra-op threads",
            "host_cores": ncpu, "torch": torch.__version__, "legs": legs,
            "streams_per_node_on_cpu": {"value": legs[2]["audio_s_per_s"], "processes": legs[2]["processes"],
                                        "note": "N independent single-thread processes (the reference's concurrency "
                                                "model) on the same window, aggregate audio-seconds per second; each "
                                                "process completes only a few steps in its 6 s budget: +-15 %"}}


def measure(w, audio, streams, beam, bbd, preroll, warmup, steps, group, mode, total, dist=None, boundary=True, kv_dtype=None,
            depth=1, exact=False):
    """one fresh batch: pre-roll (lock-step, resident audio, untimed), warm-up in the timed mode, then the timed leg"""
    sb = build_native(w, streams, beam, bbd, total, kv_dtype=kv_dtype)
    if depth > 1:
        sb.set_queue_depth(depth)
    roll(sb, audio, preroll)
    k0 = preroll + warmup
    a3 = audio.reshape(streams, -1, CHUNK)
    if mode == "strict":
        run_host(sb, step_blocks(audio, preroll, k0), np.arange(streams, dtype=np.int32))
        e, dsh, _, extra = strict_window(sb, audio, k0, steps, dist)
        out = dict(extra, elapsed=e, dec_steps_per_hop=dsh)
    else:
        nxt = np.full(streams, preroll, np.int64)
        if not boundary:
            preload_audio(sb, audio, a3.shape[1])
        if warmup > 0:
            serve(sb, a3, nxt, warmup, group, boundary=boundary, depth=depth)
        out = serve(sb, a3, nxt, steps, group, dist, boundary=boundary, depth=depth, exact=exact)
        out["next_chunk"] = nxt
    out["value"] = streams * steps * CHUNK / 16000.0 / out["elapsed"]
    return sb, out


def long_context_leg(w, streams, beam, target_T, bbd, steps, group):
    """continuous and strict throughput with every stream about `target_T` encoder frames into its utterance"""
    hops = max(2, (target_T - 24) // 16 + 2)
    preroll, warm = hops - 2, 2
    total = preroll + warm + steps + SERVED_SPARE
    audio = make_audio(streams, total, stream_offset=7000, shared=True)
    out = {"bbd": int(bbd), "steps": steps, "audio": "windows of one seeded noise buffer at per-stream offsets"}
    sb, r = measure(w, audio, streams, beam, bbd, preroll, warm, steps, group, "continuous", total)
    out.update({"value": round(r["value"], 2), "unit": "audio_s/s", "ms_per_step_equivalent": round(r["elapsed"] / steps * 1e3, 3),
                "decode_steps_per_hop": round(r["dec_steps_per_hop"], 2),
                "decode_iterations_per_step": round(r["iterations_per_step"], 2)})
    out.update(state_of(sb))
    sb.close()
    del sb
    sb, r = measure(w, audio, streams, beam, bbd, preroll, warm, steps, group, "strict", total)
    out["strict_lock_step"] = {"value": round(r["value"], 2), "ms_per_step": round(r["elapsed"] / steps * 1e3, 3),
                               "decode_steps_per_hop": round(r["dec_steps_per_hop"], 2)}
    sb.close()
    del sb, audio
    return out


def chunk_leg(w, streams, beam, bbd, chunk, group, audio_s_pre, audio_s_timed):
    """the headline's two modes at ANOTHER call size: 25 600 samples = the model's block size (2-3 encoder
    blocks and decode blocks per call), 8 192 = the reference CLI's chunk loop (speechcatcher.py:796).  Same position in the
    utterance as the headline's window (pre-roll / timed region given in audio seconds), chunk-steps/s and audio-s/s."""
    global CHUNK
    keep = CHUNK
    CHUNK = chunk
    try:
        pre = max(2, int(round(audio_s_pre * 16000.0 / chunk)) - 2)
        warm = 2
        steps = max(4, int(round(audio_s_timed * 16000.0 / chunk)))
        total = pre + warm + steps + max(4, SERVED_SPARE * 10240 // chunk)
        audio = make_audio(streams, total)
        out = {"chunk_samples": chunk, "chunk_ms": chunk * 1000 // 16000, "steps": steps,
               "window": f"chunks {pre + warm}..{pre + warm + steps - 1} of every stream ({pre} lock-step pre-roll + {warm} warm-up steps untimed)"}
        for mode in ("continuous", "strict"):
            sb, r = measure(w, audio, streams, beam, bbd, pre, warm, steps, group, mode, total)
            ent = {"value": round(r["value"], 2), "unit": "audio_s/s", "chunk_steps_per_s": round(streams * steps / r["elapsed"], 2),
                   "ms_per_step": round(r["elapsed"] / steps * 1e3, 3), "decode_steps_per_chunk": round(r["dec_steps_per_hop"], 2)}
            if mode == "continuous":
                ent["decode_iterations_per_step"] = round(r["iterations_per_step"], 2)
                out.update(ent)
                out.update(state_of(sb))
            else:
                out["strict_lock_step"] = ent
            sb.close()
            del sb
        return out
    finally:
        CHUNK = keep


def pinned_leg(argv_core, frac=8):
    """the headline leg in a child process that is confined to 1/`frac` of the host cores - what a rank gets on a node with
    `frac` GPUs (launch_ranks / the slicing under an external launcher): the open question of DESIGN section 7 is host-side
    contention of the feeder processes.  The child is started BEFORE it touches a GPU; this process only parses its line."""
    import subprocess
    cpus = sorted(os.sched_getaffinity(0))
    mine = cpus[:max(1, len(cpus) // frac)]
    env = dict(os.environ, SC_BENCH_CPUS=",".join(map(str, mine)))
    cmd = [sys.executable, os.path.abspath(__file__)] + argv_core + ["--legs", "none", "--roofline-steps", "0"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if res.returncode != 0:
        return {"error": res.stderr[-400:]}
    line = json.loads(res.stdout.strip().splitlines()[-1])
    return {"value": line["value"], "unit": "audio_s/s", "ms_per_step": line["ms_per_step"], "host_cores_used": len(mine),
            "host_cores_total": len(cpus),
            "note": f"the headline leg (same window, same mode) with the process confined to {len(mine)} of the {len(cpus)} host cores "
                    f"= a rank's share on a node with {frac} GPUs; no multi-GPU node was available: this is NOT a scaling measurement"}


_PMC_FAMILY = {"ffn_fused_kernel<256,*>": "ffn_fused_kernelILi256ELi{}ELb0ELi0",   # <256, RTT, PRO = false, WF = 0 (fp32 weights)>
               "ffn_fused_kernel<256,*,PRO>": "ffn_fused_kernelILi256ELi{}ELb1ELi0",
               "dec_layer_attn_kernel<self>": "dec_layer_attn_kernelILi256ELi32ELi10ELb1",
               "dec_layer_attn_kernel<cross>": "dec_layer_attn_kernelILi256ELi32ELi10ELb0",
               "dec_attn_flash_kernel<cross>": "dec_attn_flash_kernelILi32ELi10ELb0",
               "dec_attn_flash_kernel<self>": "dec_attn_flash_kernelILi32ELi10ELb1",
               "proj_ln_proj_kernel<256,*>": "proj_ln_proj_kernelILi256",
               "rowtile_proj_kernel<256,*>": "rowtile_proj_kernelILi256"}


ROOFLINE_WINDOW_CSV = os.path.join("profiles", "r06_bench_default_roofline_window.csv")


def pmc_traffic(kind_name):
    """HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE... as the guide's HBM section prescribes) of a kernel kind from the
    COMMITTED summary of the rocprofv3 passes of this command (tools/prof_bench.sh: kernel trace + two --pmc passes, each
    restricted to the launches between the roofline leg's sc_marker kernels, joined with this leg's own table) - a process
    cannot read the counters of its own kernels.  Returns (bytes per launch, provenance) or (None, reason)."""
    import csv
    path = os.path.join(ROOT, ROOFLINE_WINDOW_CSV)
    if not os.path.exists(path):
        return None, f"{ROOFLINE_WINDOW_CSV} not found"
    with open(path) as f:
        rows = list(csv.DictReader(r for r in f if not r.startswith("#")))
        f.seek(0)
        head = [r[1:].strip() for r in f if r.startswith("#")]
    for r in rows:
        if r["kind"] == kind_name and r.get("hbm_bytes_per_launch_pmc"):
            return int(float(r["hbm_bytes_per_launch_pmc"])), f"{ROOFLINE_WINDOW_CSV} ({'; '.join(head[:2])})"
    return None, f"{ROOFLINE_WINDOW_CSV} has no counter row for this kind"


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N rank processes
    here - fresh children of a parent that never touches a GPU (torch.cuda.device_count() does not initialise one),
    rank i bound to GPU i and to its own slice of the host cores (the reference's process pool over segments,
    speechcatcher/speechcatcher.py:474-497, is one process per worker too).  Rank 0 prints the JSON line; the exit code
    is the worst child's.  A rank that dies takes the others down instead of leaving them in a barrier."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n and os.environ.get("SC_BENCH_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {n} but only {ndev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // n)
    procs = []
    for r in range(n):
        mine = cpus[r * per:(r + 1) * per] or cpus
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SC_BENCH_CPUS=",".join(map(str, mine)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or (code if code > 0 else 1)
                for q in live:          # the exact children started above
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preroll", type=int, default=21,
                    help="untimed lock-step chunk steps in front of the warm-up: the timed window then sits in the middle of "
                         "SURVEY 8(d)'s 60 s utterance (hops 26-45 with --warmup 5 --steps 20) instead of at its cheap beginning")
    ap.add_argument("--mode", choices=["continuous", "strict"], default="continuous",
                    help="continuous: every stream is served on its own through sc_submit / sc_poll (continuous batching: a "
                         "reply is delivered when ITS blocks are decoded - the reference's concurrency model, one independent "
                         "call loop per stream); strict: one batched sc_push per chunk step, every stream waits for the slowest "
                         "one of the batch.  Per call the results are the same.  The other mode is reported as an extra key")
    ap.add_argument("--streams", type=int, default=128, help="streams per GPU")
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--bbd", type=int, default=0, help="block boundary detection (reference CLI default: on)")
    ap.add_argument("--chunk", type=int, default=10240,
                    help="samples per chunk step (10240 = 640 ms = one encoder hop; also 8192 = CLI default, 25600 = block size)")
    ap.add_argument("--queue-depth", type=int, default=1,
                    help="continuous mode: chunks a stream keeps with the engine (sc_streams_set_queue_depth); 1 = call -> reply -> "
                         "next call (the headline), 2+ = the next chunk is submitted while the previous one decodes (file / backlog hosts)")
    ap.add_argument("--encoder-batch", type=int, default=0,
                    help="continuous mode: merged encoder groups are issued when they hold this many streams "
                         "(sc_streams_set_encoder_batch; 0 = the engine's default)")
    ap.add_argument("--legs", choices=["core", "all", "none"], default="core",
                    help="core (default, <= 90 s): the headline + its roofline, strict lock-step, exact-steps, the 25 600- and 8 192-sample "
                         "call sizes, single stream, fp16 mode, 1/8 of the host cores, CPU baseline; all: also resident audio, fp16 K|V, "
                         "split16, block-boundary detection on, queue depth 2, long context (T ~ 1000 / 4500); none: the headline only")
    ap.add_argument("--roofline-csv", default="",
                    help="write the per-kernel table of the roofline leg (launches, avg us, algorithmic bytes / flops per launch) here; "
                         "tools/prof_bench.sh joins it with the rocprofv3 passes of the SAME launches (sc_marker brackets the leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the leg of the mode that is not the headline")
    ap.add_argument("--no-resident", action="store_true", help="skip the resident-audio / no-read-back leg")
    ap.add_argument("--no-long-context", action="store_true", help="skip the T ~ 1000 / T ~ 4500 legs")
    ap.add_argument("--poll-group", type=int, default=0,
                    help="continuous: sc_poll returns when at least this many replies are ready (0: streams / 8)")
    ap.add_argument("--roofline-steps", type=int, default=2,
                    help="extra (untimed for `value`) steps with per-launch HIP-event timing of the hot kernels")
    ap.add_argument("--kv-dtype", choices=["float32", "float16"], default="float32",
                    help="float16: self-/cross-attention K|V caches stored in fp16, arithmetic fp32 (opt-in; results "
                         "differ from the fp32 reference within the tolerance stated in tests/test_gpu_native.py)")
    ap.add_argument("--ffn-dtype", choices=["float32", "float16", "split16"], default="float32",
                    help="float16: feed-forward weights (encoder and decoder) and the encoder's attention projections in fp16 "
                         "with fp16 MFMA inputs, fp32 accumulation (opt-in, BASELINE configs[4]; never the headline).  "
                         "split16: the feed-forward kernels evaluate their fp32 product sums on the fp16 matrix pipe from "
                         "fp16 hi + lo splits of both operands (three MFMAs per sum, fp32-grade results: sc_ffn_ln_s)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if os.environ.get("SC_BENCH_CPUS"):      # a rank started by launch_ranks: its own slice of the host cores
        mine = [int(c) for c in os.environ["SC_BENCH_CPUS"].split(",")]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(8, len(mine))))
    elif int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) > 1:
        # ranks started by an external launcher (torch.distributed.run): the same slicing - rank i of the node feeds its GPU
        # from its own contiguous share of the allowed cores instead of migrating over all of them beside 7 other feeders
        try:
            nloc = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"]))
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            cpus = sorted(os.sched_getaffinity(0))
            per = len(cpus) // nloc
            if per >= 2:
                os.sched_setaffinity(0, cpus[lr * per:(lr + 1) * per])
                torch.set_num_threads(max(1, min(8, per)))
        except (OSError, ValueError, KeyError):
            pass
    if args.legs == "none":
        args.no_cpu_baseline = args.no_single_stream = args.no_other_mode = args.no_resident = args.no_long_context = True
    extended = args.legs == "all"
    global CHUNK, KV_DTYPE, FFN_DTYPE, ENCODER_BATCH
    ENCODER_BATCH = args.encoder_batch
    CHUNK = args.chunk
    KV_DTYPE = args.kv_dtype
    FFN_DTYPE = args.ffn_dtype
    group = args.poll_group or max(1, args.streams // 8)
    S = args.streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SC_BENCH_SINGLE_DEVICE") == "1":    # multi-rank control flow on a 1-GPU box (tests)
        local_rank = 0
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)          # BEFORE init_process_group: RCCL binds the rank to this device
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  SC_DIST_BACKEND=gloo + SC_BENCH_SINGLE_DEVICE=1 lets the multi-rank control flow
        # be smoke-tested on a 1-GPU box (collectives on CPU tensors).
        backend = os.environ.get("SC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_device = device if (dist is None or dist.get_backend() == "nccl") else "cpu"
    per_rank_elapsed = None

    k0 = args.preroll + args.warmup
    total_steps = k0 + args.steps + args.roofline_steps + SERVED_SPARE
    w = make_weights(device)
    audio = make_audio(S, total_steps, stream_offset=rank * S)
    a3 = audio.reshape(S, -1, CHUNK)
    sb, head = measure(w, audio, S, args.beam, bool(args.bbd), args.preroll, args.warmup, args.steps, group, args.mode,
                       total_steps, dist, depth=args.queue_depth)
    lib = sb.lib
    elapsed, dec_steps_per_hop = head["elapsed"], head["dec_steps_per_hop"]
    state = state_of(sb)
    kv_rows = sb.kv_rows

    # Roofline leg: the SAME workload continues in the SAME mode for a few more steps with hipGraph replay switched
    # off, so that every launch of the hot kernels can be bracketed by HIP events on its launch stream.
    NK = 13   # scasr.h: SC_PROF_KINDS
    ms, fl, by = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_double * NK)()
    nn = (C.c_longlong * NK)()
    ev_over_ms, xattn_bytes, xattn_flops = 0.0, {}, {}
    if args.roofline_steps > 0:
        sb.set_graphs(False)
        sb.take_attn_counters()
        torch.cuda.synchronize()
        lib.sc_marker(0, sb.hip_stream)     # brackets THIS leg in a rocprofv3 trace / counter pass (tools/prof_bench.sh)
        torch.cuda.synchronize()
        lib.sc_prof_enable(1)
        if args.mode == "strict":
            run_host(sb, step_blocks(audio, k0 + args.steps, k0 + args.steps + args.roofline_steps), np.arange(S, dtype=np.int32))
        else:
            # (the per-launch timing stops with the last timed reply: the untimed drain of the stragglers runs at small
            # compaction buckets that the steady state never sees)
            rows_at_target = []

            def stop_timing(counters):
                lib.sc_prof_enable(0)
                lib.sc_marker(1, sb.hip_stream)     # (the device is idle here: serve() synchronises before it calls back)
                rows_at_target.append(counters)

            serve(sb, a3, head["next_chunk"], args.roofline_steps, group, at_target=stop_timing, depth=args.queue_depth)
        torch.cuda.synchronize()
        lib.sc_prof_enable(0)
        if args.mode == "strict":
            lib.sc_marker(1, sb.hip_stream)
        torch.cuda.synchronize()
        lib.sc_prof_collect_kinds(ms, fl, by, nn, NK)
        sb.set_graphs(True)
        rows = sb.take_attn_counters()
        if args.mode != "strict":
            rows = rows_at_target[0]
        ev_over_ms = float(lib.sc_prof_event_overhead_ms(sb.hip_stream))
        # algorithmic K|V bytes: the cross-attention reads the T encoder rows of every active stream once per layer and
        # step; the self-attention the DISTINCT (position, slot) rows its hypotheses descend from (device counter) -
        # a row = K|V of all heads = 2d elements
        esz = 2 if KV_DTYPE == "float16" else 4
        rowb = 2 * XL.d_model * esz
        # (kind 12, the stream-resident layer kernel of batches > 128 streams: both attentions of a layer in one launch)
        xattn_bytes = {7: float(rows["cross_rows"][0]) * rowb, 11: float(rows["cross_rows"][1]) * rowb,
                       6: float(rows["self_distinct_rows"][0]) * rowb, 10: float(rows["self_distinct_rows"][1]) * rowb,
                       12: float(rows["cross_rows"][2] + rows["self_distinct_rows"][2]) * rowb}
        # ... and the attention's own flops: q.k + p.v = 4 d per (hypothesis, key, layer) - keys = the T frames / the L tokens
        fa = 4.0 * XL.d_model * args.beam
        xattn_flops = {7: fa * rows["cross_rows"][0], 11: fa * rows["cross_rows"][1],
                       6: fa * rows["self_positions"][0], 10: fa * rows["self_positions"][1],
                       12: fa * (rows["cross_rows"][2] + rows["self_positions"][2])}

    if dist is not None:
        t = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        each = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(each, t)
        per_rank_elapsed = [float(x.item()) for x in each]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the path's single collective: gather of the final text (token ids + positions + length + score)
        from speechcatcher_amd.distributed import gather_final_hypotheses, pack_hypotheses
        a = sb.hypotheses_arrays(list(range(S)), nbest=1)
        n1 = a["lens"][:, 0]
        payload = pack_hypotheses([a["ids"][i, 0, :n1[i]] for i in range(S)], [a["xpos"][i, 0, :n1[i]] for i in range(S)],
                                  a["score"][:, 0], 640, coll_device)
        gathered = gather_final_hypotheses(payload, S)
        assert len(gathered) == world and len(gathered[0]) == S

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    audio_s = world * S * args.steps * CHUNK / 16000.0
    value = audio_s / elapsed
    default_workload = (S == 128 and CHUNK == 10240 and args.beam == 10 and not args.bbd and args.mode == "continuous" and
                        KV_DTYPE == "float32" and FFN_DTYPE == "float32" and args.queue_depth == 1)
    # dominant kernel of the path = the kernel kind with the largest summed launch time in the roofline leg
    names = ["gemm_naive_kernel", "gemm_skinny_kernel", "gemm_mfma_kernel<128,128>", "gemm_mfma_kernel<64,64>",
             "proj_ln_proj_kernel<256,*>", "ffn_fused_kernel<256,*>", "dec_attn_flash_kernel<self> (stand-alone decoder self-attention: six-launch layers)",
             "dec_attn_flash_kernel<cross> (stand-alone decoder cross-attention: six-launch layers)", "rowtile_proj_kernel<256,*>",
             "ffn_fused_kernel<256,*,PRO> (decoder layer FFN: reduce of the head partials + norm3 prologue)",
             "dec_layer_attn_kernel<self> (decoder layer: reduce + norm1 + Q|K|V + self-attention + out-projection; 1 or 4 heads per workgroup)",
             "dec_layer_attn_kernel<cross> (decoder layer: reduce + norm2 + q + cross-attention + out-projection; 1 or 4 heads per workgroup)",
             "dec_layer_stream_kernel (stream-resident decoder layer: both attentions of a layer for all heads, one workgroup per stream; batches > 128 streams)"]
    net = [max(ms[i] - nn[i] * ev_over_ms, 0.0) for i in range(NK)]
    tot_ms = max(sum(net), 1e-9)
    # every kind against BOTH roofs: algorithmic flops (incl. the attention's q.k / p.v) at the f32 matrix peak, algorithmic
    # bytes (operands once + the K|V rows the attention must read) at the HBM peak; the bound is the one that takes longer
    per_kernel, bound_of = [], {}
    for i in range(NK):
        if nn[i] == 0:
            continue
        t_s = max(net[i], 1e-9) * 1e-3
        flops_i = fl[i] + xattn_flops.get(i, 0.0)
        bytes_i = by[i] + xattn_bytes.get(i, 0.0)
        tf, gbs = flops_i / t_s / 1e12, bytes_i / t_s / 1e9
        f_mfma, f_hbm = tf / PEAK_F32_MFMA_TFLOPS, gbs / PEAK_HBM_GBS
        bound_of[i] = ("mfma", tf, f_mfma, flops_i, bytes_i) if f_mfma >= f_hbm else ("hbm", gbs, f_hbm, flops_i, bytes_i)
        ent = {"kernel": names[i], "launches": int(nn[i]), "avg_launch_us": round(net[i] * 1e3 / nn[i], 2),
               "share_of_timed_kernel_time": round(net[i] / tot_ms, 4), "bound": bound_of[i][0],
               "frac_of_bound": round(bound_of[i][2], 4)}
        if flops_i > 0:
            ent["tflops"] = round(tf, 3)
            ent["frac_of_f32_mfma_peak"] = round(f_mfma, 4)
        if bytes_i > 0:
            ent["hbm_gbs_algorithmic"] = round(gbs, 1)
            ent["frac_of_hbm_peak"] = round(f_hbm, 4)
        if xattn_bytes.get(i, 0) > 0:
            ent["kv_bytes_per_launch_avg"] = int(xattn_bytes[i] / nn[i])
        per_kernel.append(ent)
    roof = None
    if any(nn[i] > 0 for i in range(NK)):
        v = max(range(NK), key=lambda i: net[i])
        raw_us = ms[v] * 1e3 / nn[v]
        t_ms = max(net[v], 1e-9)
        kind, ach, frac, flops_v, bytes_v = bound_of[v]
        traffic_v, traffic_src = pmc_traffic(names[v].split(" (")[0]) if default_workload else (None, "not the default workload")
        # traffic: HBM bytes per launch need rocprofv3 --pmc passes of this command (a process cannot read the
        # counters of its own kernels): `traffic` is taken from the committed summary of those passes
        # (profiles/r04_bench_default_pmc_hbm_traffic.csv, tools/prof_bench.sh: FETCH_SIZE x2 + WRITE_SIZE per launch,
        # launch-weighted over the kernel family) when the file is there and this is the default workload, else null.
        roof = {"bound": kind, "achieved": round(ach, 3 if kind == "mfma" else 1),
                "peak": PEAK_F32_MFMA_TFLOPS if kind == "mfma" else PEAK_HBM_GBS, "unit": "TFLOP/s" if kind == "mfma" else "GB/s",
                "frac": round(frac, 4),
                "flops_per_launch_avg": round(flops_v / nn[v] / 1e6, 1),
                "flops_unit": "MFLOP algorithmic per launch (DESIGN.md section 4)",
                "algorithmic_bytes_per_launch_avg": int(bytes_v / nn[v]),
                "kernel": names[v], "traffic": traffic_v, "traffic_source": traffic_src,
                "avg_launch_us": round(t_ms * 1e3 / nn[v], 2), "avg_launch_us_raw_events": round(raw_us, 2),
                "event_pair_overhead_us": round(ev_over_ms * 1e3, 2), "launches_timed": int(nn[v]),
                "share_of_timed_kernel_time": round(net[v] / tot_ms, 4),
                "measured_over": f"{args.roofline_steps} steps following the timed region, same workload, "
                                 "hipGraph replay off, HIP events around every launch",
                "per_kernel": per_kernel}

    if args.roofline_csv and roof is not None:
        import csv
        with open(args.roofline_csv, "w", newline="") as f:
            f.write(f"# roofline leg of `python bench.py` ({args.roofline_steps} steps behind the timed region, hipGraph replay off, HIP events "
                    "around every launch); launches between sc_marker_kernel<0> and <1>\n")
            wr = csv.writer(f)
            wr.writerow(["kind", "launches", "avg_launch_us_events", "algorithmic_bytes_per_launch", "algorithmic_mflop_per_launch"])
            for i in range(NK):
                if nn[i]:
                    wr.writerow([names[i].split(" (")[0], int(nn[i]), round(net[i] * 1e3 / nn[i], 2),
                                 int((by[i] + xattn_bytes.get(i, 0.0)) / nn[i]), round((fl[i] + xattn_flops.get(i, 0.0)) / nn[i] / 1e6, 2)])

    # whole chunk step against the matrix-core peak: algorithmic FLOPs of SURVEY 8(d) / wall time
    gflop_step = args.streams * (GFLOP_ENCODER_SIDE_PER_HOP * CHUNK / 10240.0 + GFLOP_PER_DECODE_STEP * args.beam / 10.0
                                 * dec_steps_per_hop)
    ms_step = elapsed / args.steps * 1e3
    whole = {"algorithmic_gflop_per_step": round(gflop_step, 1),
             "achieved_tflops_whole_step": round(gflop_step / ms_step, 2),
             "frac_of_f32_mfma_peak": round(gflop_step / ms_step / PEAK_F32_MFMA_TFLOPS, 4),
             "note": f"{args.streams} x (3.83 + 0.437 x decode steps per hop) GFLOP per chunk step (SURVEY 8(d)) / ms_per_step; "
                     "attention and the CTC scan over the prefix are not in this FLOP count (see with_attention / hbm)"}
    if head.get("attn"):
        c = XL
        d, F, V, Ld, Le, W = c.d_model, c.ffn_dim, c.vocab_size, c.dec_layers, c.enc_layers, args.beam
        at = head["attn"]
        cross_rows, self_pos, self_rows = sum(at["cross_rows"]), sum(at["self_positions"]), sum(at["self_distinct_rows"])
        # decoder attention: q.k and p.v of every hypothesis over its L tokens / the T frames: 4 d flop per key, layer, hypothesis
        gflop_attn = 4.0 * d * W * (cross_rows + self_pos) / 1e9 / args.steps
        esz = 2 if KV_DTYPE == "float16" else 4
        K = 40
        sum_T = cross_rows / float(Ld)                   # sum over decode iterations and active streams of T
        hbm = {"decoder_weights_per_iteration": head["iterations"] * 4.0 * (Ld * (6 * d * d + 2 * d * F) + V * d),
               "encoder_side_weights_per_group": head["encoder_groups"] * 4.0 * (Le * (4 * d * d + 2 * d * F) + 9 * d * d
                                                                                  + c.conv_freq2 * d * d + V * d + Ld * 2 * d * d),
               "cross_attention_kv": cross_rows * 2.0 * d * esz,
               "self_attention_kv_distinct_rows": self_rows * 2.0 * d * esz,
               # prefix scan per (iteration, stream, frame): W*K table elements read, r of the W hypotheses read, r of the
               # W*K candidates written at every 16th frame (checkpoints); rebuild of the W winners: their table column
               # + r of the prefix (2 W) read, checkpoints read, r written (2 W)
               "ctc_table_and_state": sum_T * 4.0 * (W * K + 2 * W + 2 * W * K / 16.0 + 7 * W)}
        tot = sum(hbm.values())
        whole["with_attention"] = {"algorithmic_gflop_per_step": round(gflop_step + gflop_attn, 1),
                                   "attention_gflop_per_step": round(gflop_attn, 1),
                                   "achieved_tflops": round((gflop_step + gflop_attn) / ms_step, 2),
                                   "frac_of_f32_mfma_peak": round((gflop_step + gflop_attn) / ms_step / PEAK_F32_MFMA_TFLOPS, 4)}
        whole["hbm"] = {"algorithmic_gb_per_step": {k: round(v / 1e9 / args.steps, 3) for k, v in hbm.items()},
                        "total_gb_per_step": round(tot / 1e9 / args.steps, 3),
                        "achieved_gbs": round(tot / 1e9 / elapsed, 1), "peak_gbs": PEAK_HBM_GBS,
                        "frac_of_hbm_peak": round(tot / 1e9 / elapsed / PEAK_HBM_GBS, 4),
                        "decode_iterations": int(head["iterations"]), "encoder_groups": int(head["encoder_groups"]),
                        "self_attention_distinct_rows_per_position": round(self_rows / max(self_pos, 1), 3),
                        "note": "algorithmic HBM bytes of the timed window / its wall time: weights once per decode iteration "
                                "(one graph replay = one pass over the decoder) and once per encoder group, K|V rows the "
                                "attention has to read (cross: T rows per active stream, layer and step; self: the distinct "
                                "rows, device counter), CTC table + forward variables of the prefix scan; activations "
                                "(< 3 %) left out.  The weights mostly stay in the 256 MB Infinity Cache between iterations: "
                                "this is an upper bound of the HBM traffic the algorithm needs, not a PMC reading"}
    sb.close()
    del sb

    def leg(mode, boundary=True, kv_dtype=None, weights=None, bbd=None, depth=1):
        sbx, r = measure(weights or w, audio, S, args.beam, bool(args.bbd) if bbd is None else bbd, args.preroll, args.warmup, args.steps, group, mode, total_steps,
                         boundary=boundary, kv_dtype=kv_dtype, depth=depth)
        sbx.close()
        o = {"value": round(r["value"], 2), "unit": "audio_s/s", "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 3),
             "decode_steps_per_hop": round(r["dec_steps_per_hop"], 2)}
        if mode == "continuous":
            o.update({"decode_iterations_per_step": round(r["iterations_per_step"], 2), "polls_per_step": round(r["polls_per_step"], 2),
                      "chunks_per_stream_min_max": r["chunks_per_stream_min_max"]})
        return o

    resident = None
    if extended and not args.no_resident and world == 1 and args.mode == "continuous":
        resident = leg("continuous", boundary=False)
        resident["headline_over_this"] = round(value / resident["value"], 4)
        resident["note"] = "same window and mode, audio resident in HBM (sc_submit with NULL pcm), no hypothesis read-back"

    other = None
    if not args.no_other_mode and world == 1:
        om = "strict" if args.mode == "continuous" else "continuous"
        other = leg(om)
        other["headline_over_this"] = round(value / other["value"], 4)
        other["note"] = ("STRICT lock-step: one batched sc_push per chunk step (host PCM in, hypotheses out), every block "
                         "completes inside its call, every stream waits for the slowest stream of the batch" if om == "strict" else
                         "continuous batching through sc_submit / sc_poll, same window")

    exact = None
    if not args.no_other_mode and world == 1 and args.mode == "continuous":
        sbx, r = measure(w, audio, S, args.beam, bool(args.bbd), args.preroll, args.warmup, args.steps, group, "continuous", total_steps,
                         depth=args.queue_depth, exact=True)
        sbx.close()
        del sbx
        exact = {"value": round(r["value"], 2), "unit": "audio_s/s", "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 3),
                 "chunks_per_stream_min_max": r["chunks_per_stream_min_max"], "headline_over_this": round(value / r["value"], 4),
                 "note": "the same leg with every stream served EXACTLY `steps` chunks (a stream that has had its share is not fed "
                         "again; the clock stops with the last reply): in the headline's closed loop the clock stops after S x steps "
                         "replies whoever sent them, so streams whose blocks need fewer decode steps contribute more chunks "
                         "(chunks_per_stream_min_max of `continuous`) - this is the equal-weight variant; its tail runs at thinning "
                         "buckets like strict lock-step does in every step"}

    chunk_legs = None
    if not args.no_other_mode and world == 1 and CHUNK == 10240 and KV_DTYPE == "float32" and FFN_DTYPE == "float32":
        chunk_legs = {}
        for ch in (25600, 8192):
            try:
                chunk_legs[f"chunk_{ch}"] = chunk_leg(w, S, args.beam, bool(args.bbd), ch, group, k0 * 0.64, args.steps * 0.64)
            except Exception as e:  # noqa: BLE001
                chunk_legs[f"chunk_{ch}"] = {"error": repr(e)}

    pinned = None
    if not args.no_other_mode and world == 1 and "SC_BENCH_CPUS" not in os.environ:
        try:
            core = ["--streams", str(S), "--steps", str(args.steps), "--warmup", str(args.warmup), "--preroll", str(args.preroll),
                    "--beam", str(args.beam), "--bbd", str(args.bbd), "--chunk", str(CHUNK), "--mode", args.mode,
                    "--kv-dtype", KV_DTYPE, "--ffn-dtype", FFN_DTYPE, "--queue-depth", str(args.queue_depth),
                    "--encoder-batch", str(args.encoder_batch), "--poll-group", str(args.poll_group)]
            pinned = pinned_leg(core)
            if "value" in pinned:
                pinned["over_headline"] = round(pinned["value"] / value, 4)
        except Exception as e:  # noqa: BLE001
            pinned = {"error": repr(e)}

    kv16 = None
    if extended and not args.no_other_mode and world == 1 and KV_DTYPE == "float32":
        kv16 = leg(args.mode, kv_dtype="float16")
        kv16["over_headline"] = round(kv16["value"] / value, 4)
        kv16["note"] = ("NOT the headline: the same leg with the self- / cross-attention K|V caches STORED in fp16 (arithmetic, softmax and "
                        "all scores fp32; opt-in `kv_dtype`): tools/fp16_mode_stats.py - 256 streams x 7 chunks, no hypothesis of any "
                        "stream changes, best scores within 7e-5 of the fp32 run")

    fp16_mode = None
    if not args.no_other_mode and world == 1 and KV_DTYPE == "float32" and FFN_DTYPE == "float32":
        w_half = make_weights(device, "float16")
        fp16_mode = leg(args.mode, weights=w_half, kv_dtype="float16")
        del w_half
        fp16_mode["over_headline"] = round(fp16_mode["value"] / value, 4)
        fp16_mode["dtype"] = ("fp16 weights + fp16 MFMA inputs (feed-forward, encoder and decoder attention projections), fp16 K|V "
                              "caches, fp16 partial products between the decoder's kernels; fp32 accumulation, LayerNorm, softmax, "
                              "output layer, log-softmax, CTC scan and scores")
        fp16_mode["note"] = ("NOT the headline: BASELINE configs[4]'s mode (`--ffn-dtype float16 --kv-dtype float16`).  No fp16 run of the "
                             "reference's native decoder exists (speechcatcher.py:205-210 disables it): parity for this mode is the ids of the "
                             "six XL fixtures + a bound on the share of streams whose best hypothesis moves (tests/test_gpu_baseline_size.py)")

    # wider batches on the one GPU (round 6): 256 streams in fp32 - the regime of the stream-resident decoder layers (one workgroup
    # per stream: a 256-stream bucket is ONE round of the 256 compute units) - and, with --legs all, the per-GPU share of BASELINE
    # configs[4] (256 streams, fp16 mode).  Same window, same closed loop, their own audio.
    wide = None
    if not args.no_other_mode and world == 1 and S == 128 and CHUNK == 10240 and KV_DTYPE == "float32" and FFN_DTYPE == "float32":
        wide = {}
        S2 = 256
        audio2 = make_audio(S2, total_steps)
        steps2 = max(8, args.steps // 2)
        for name, wdt, kvd in (("streams_256_f32", None, None),) + ((("config4_share_256_fp16", "float16", "float16"),) if extended else ()):
            try:
                wx = make_weights(device, wdt) if wdt else w
                sbx, r = measure(wx, audio2, S2, args.beam, bool(args.bbd), args.preroll, args.warmup, steps2, S2 // 8, args.mode, total_steps,
                                 kv_dtype=kvd, depth=args.queue_depth)
                kvr = sbx.kv_rows
                sbx.close()
                del sbx
                if wdt:
                    del wx
                wide[name] = {"value": round(r["value"], 2), "unit": "audio_s/s", "streams": S2, "steps": steps2,
                              "ms_per_step": round(r["elapsed"] / steps2 * 1e3, 3), "over_headline": round(r["value"] / value, 4),
                              "decode_iterations_per_step": round(r.get("iterations_per_step", 0.0), 2),
                              "self_attention_kv_pool_rows_per_stream_and_layer": kvr}
            except Exception as e:  # noqa: BLE001
                wide[name] = {"error": repr(e)}
        del audio2
        wide["note"] = ("NOT the headline (BASELINE configs[2] names 128 streams per GPU): the same closed loop with 256 streams on the one GPU.  "
                        "fp32: buckets of more than 128 streams take the stream-resident decoder layers (csrc/decoder_stream.hip, two launches "
                        "per layer, bit-identical results).  config4_share (--legs all): 256 streams in the fp16 mode = one GPU's share of "
                        "BASELINE configs[4]")

    bbd_on = None
    if extended and not args.no_other_mode and world == 1 and not args.bbd:
        bbd_on = leg(args.mode, bbd=True)
        bbd_on["over_headline"] = round(bbd_on["value"] / value, 4)
        bbd_on["note"] = ("the other search regime of BASELINE.md section 3: the same leg WITH block-boundary detection (the reference CLI's "
                          "default).  On random weights it ends most blocks after ~1.3 decode steps (real speech: 3-5), so this is an "
                          "upper bound for a real checkpoint as the headline (detection off, ~9 steps per block) is a lower bound")

    l_like = None
    if extended and not args.no_other_mode and world == 1 and KV_DTYPE == "float32" and FFN_DTYPE == "float32":
        from speechcatcher_amd.config import L_LIKE
        w_l = make_weights(device, cfg=L_LIKE)
        l_like = leg(args.mode, weights=w_l)
        del w_l
        l_like["note"] = ("NOT the headline: the per-GPU share of BASELINE configs[3] (`en_streaming_transformer_l`, 128 streams per GPU) on "
                          "ASSUMED stand-in dims (speechcatcher_amd/config.py: L_LIKE = d 256, 4 heads of 64, 18 + 8 blocks; the real "
                          "config.yaml is not available offline).  Head dim 64 runs the head-parallel layer kernels since round 6 "
                          "(three launches per layer, one head per workgroup; rounds 1-5: the six-launch form, 5798 audio-s/s)")
    queued = None
    if extended and not args.no_other_mode and world == 1 and args.mode == "continuous" and args.queue_depth == 1:
        queued = leg("continuous", depth=2)
        queued["over_headline"] = round(queued["value"] / value, 4)
        queued["note"] = ("NOT the headline: the same leg with TWO chunks per stream at the engine (sc_streams_set_queue_depth(2)): a host "
                          "that has the audio already - a file (the reference CLI's chunk loop, speechcatcher.py:574-592), a backlog - "
                          "submits a stream's next chunk while the previous one is still decoding, so its frontend + encoder run beside "
                          "the decoding and the stream does not idle between its reply and its next call.  Per chunk the replies are "
                          "those of the one-at-a-time protocol (tests/test_gpu_native.py test_native_queued_chunks_*).  Live streams "
                          "deliver a chunk per 640 ms and never have a second one ready: for them tools/realtime_sim.py is the measure.  "
                          "At 128 streams the decode iterations are work-bound, so fewer but fuller iterations buy ~1 %; the gain is at "
                          "fewer streams (32 streams: +12 %, one stream: +6 %; DESIGN section 4)")

    split16 = None
    if extended and not args.no_other_mode and world == 1 and KV_DTYPE == "float32" and FFN_DTYPE == "float32":
        w_split = make_weights(device, "split16")
        split16 = leg(args.mode, weights=w_split)
        del w_split
        split16["over_headline"] = round(split16["value"] / value, 4)
        split16["note"] = ("NOT the headline: the same leg with `ffn_dtype = proj_dtype = \"split16\"` - the fused feed-forward kernels of all "
                           "encoder and decoder layers and the encoder's attention projections evaluate their fp32 product sums on the fp16 "
                           "matrix pipe: both operands split into fp16 hi + lo / 2^11, three v_mfma_f32_16x16x32_f16 per sum instead of "
                           "eight v_mfma_f32_16x16x4_f32, fp32 accumulation (sc_ffn_ln_s / sc_rowtile_proj_s; 22-bit products: error "
                           "against float64 within 4x the fp32 kernel's, tests/test_gpu_ops.py).  "
                           "Holds the fp32 parity bar unrelaxed: all hypotheses of the six XL reference fixtures on both engines "
                           "(tests/test_gpu_native.py), and on 256 streams x 7 chunks no hypothesis of any beam differs from the fp32 "
                           "run, best scores within 1.5e-5 (tools/fp16_mode_stats.py split16)")

    single = None
    if not args.no_single_stream and world == 1:
        sb1, r1 = measure(w, audio[:1], 1, args.beam, bool(args.bbd), args.preroll, args.warmup, args.steps, 1, "strict", total_steps)
        sb1.close()
        hop_s = CHUNK / 16000.0
        single = {"ms_per_hop": round(r1["elapsed"] / args.steps * 1e3, 3), "rtf": round(r1["elapsed"] / (args.steps * hop_s), 5),
                  "x_realtime": round(args.steps * hop_s / r1["elapsed"], 1)}
    del audio, a3

    long_ctx = None
    if extended and not args.no_long_context and world == 1 and CHUNK == 10240:
        long_ctx = {}
        try:
            long_ctx["T1000"] = long_context_leg(w, S, args.beam, 1000, False, 8, group)
            long_ctx["T4500"] = long_context_leg(w, S, args.beam, 4500, True, 6, group)
            long_ctx["note"] = ("T1000: search without block-boundary detection like the headline (hypotheses near the "
                                "reference's 500-step bound); T4500 = a 180 s CLI segment, with block-boundary detection "
                                "(the reference CLI's default): without it every stream has hit the 500-step bound by then")
        except Exception as e:  # noqa: BLE001
            long_ctx["error"] = repr(e)

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(k0, beam=args.beam, bbd=bool(args.bbd))
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}

    semantics = ("continuous batching: every stream is called, answers and is called again on its own (sc_submit / sc_poll) - a "
                 "reply is delivered when ITS decode blocks are done, " + ("exactly one chunk per stream outstanding (the reference's "
                 "concurrency model: one independent call loop per stream, speechcatcher_server.py:331-397)" if args.queue_depth == 1 else
                 f"{args.queue_depth} chunks per stream at the engine (the next chunk is submitted while the previous one decodes: a "
                 "host that has the audio already)") + "; per call the "
                 "results are those of the strict lock-step run (same blocks, steps and kernels; a stream's fp32 summation order follows the kernel form of its bucket, DESIGN.md section 2)" if args.mode == "continuous" else
                 "strict lock-step: one batched call per chunk step, every block completes inside its call")
    out = {
        "metric": f"concurrent real-time streams (audio-seconds/s), de_xl dims, {CHUNK * 1000 // 16000} ms ({CHUNK}-sample) chunk steps, beam 10 CTC+attention",
        "value": round(value, 2), "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32" if KV_DTYPE == "float32" and FFN_DTYPE == "float32" else
                  "f32 except: " + ", ".join(x for x in (("fp16 K|V caches" if KV_DTYPE != "float32" else ""),
                                                         ("fp16 weights and MFMA inputs in the feed-forward, the encoder's and (with fp16 K|V caches) the decoder's "
                                                          "attention projections, fp16 partial products between the decoder's kernels "
                                                          "(fp32 accumulation)" if FFN_DTYPE == "float16" else ""),
                                                         ("feed-forward and encoder attention-projection product sums from fp16 hi + lo splits of both "
                                                          "fp32 operands on the fp16 matrix pipe (three MFMAs per sum, fp32 accumulation; ~2^-22 "
                                                          "relative per product)"
                                                          if FFN_DTYPE == "split16" else "")) if x)),
        "data": "synthetic",
        "config": {"workload": f"de_streaming_transformer_xl dims, {S} concurrent synthetic streams/GPU "
                               f"(batched encoder + batched beam), beam {args.beam}, chunk {CHUNK} samples, bbd {args.bbd}",
                   "streams_per_gpu": S, "chunk_samples": CHUNK, "beam": args.beam, "bbd": args.bbd,
                   "mode": args.mode, "queue_depth": args.queue_depth, "semantics": semantics,
                   "self_attention_kv_pool_rows_per_stream_and_layer": kv_rows,
                   "step": (f"one step = one {CHUNK}-sample chunk of EVERY stream = {S} calls / replies; the clock stops when "
                            f"{S} x steps replies have been delivered" if args.mode == "continuous" else
                            f"one step = one batched call with a {CHUNK}-sample chunk of every stream"),
                   "boundary": "inside the timed region: host PCM chunks in (pinned staging, one H2D per admission), best "
                               "hypothesis of every answering stream out (token ids + positions + scores, one D2H per reply batch)",
                   "window": f"chunks {k0}.. of every stream ({args.preroll} lock-step pre-roll + {args.warmup} warm-up steps untimed), "
                             f"at the end of the window: T = {state['encoder_frames_T'][0]}..{state['encoder_frames_T'][1]} encoder "
                             f"frames, {state['tokens_L'][0]}..{state['tokens_L'][1]} tokens per hypothesis",
                   "engine": "C++ (sc_submit / sc_poll / sc_get_hyps_batch, csrc/streams.hip)" if args.mode == "continuous" else
                             "C++ (sc_push + sc_get_hyps_batch, csrc/streams.hip)",
                   "parallelism": f"streams sharded x{world}, no collective in the hot loop"},
        "chunk_steps_per_s": round(world * S * args.steps / elapsed, 2),
        "per_rank": (None if per_rank_elapsed is None else
                     [{"rank": r, "gpu": r, "audio_s_per_s": round(S * args.steps * CHUNK / 16000.0 / e, 2),
                       "ms_per_step": round(e / args.steps * 1e3, 3)} for r, e in enumerate(per_rank_elapsed)]),
        "decode_steps_per_hop": round(dec_steps_per_hop, 2),
        "whole_step": whole, "roofline": roof, "cpu_baseline": cpu, "single_stream": single,
        "resident_no_readback": resident, ("strict_lock_step" if args.mode == "continuous" else "continuous"): other,
        "value_exact_steps": (exact or {}).get("value"),
        "exact_steps": exact, "chunk_sizes": chunk_legs, "one_eighth_of_host_cores": pinned, "wider_batches": wide,
        "kv_cache_fp16": kv16, "fp16_mode": fp16_mode, "ffn_split16": split16, "bbd_on": bbd_on, "queue_depth_2": queued, "l_like_dims": l_like, "long_context": long_ctx,
        "legs": args.legs,
    }
    if args.mode == "continuous":
        out["continuous"] = {k: head[k] for k in ("iterations_per_step", "polls_per_step", "chunks_per_stream_min_max")}
        out["continuous"]["poll_min_done"] = group
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
