#!/usr/bin/env python3
"""Benchmark of the streaming-ASR hot path on MI355X.

One "step" = one chunk step (10 240 samples = 640 ms of 16 kHz audio) of the
whole path - log-mel frontend, Conv2d subsampling, contextual-block encoder,
blockwise-synchronous beam search (decoder + CTC prefix scorer) - for EVERY
stream of the batch (S streams per GPU, default 128 = BASELINE.json
configs[2]; `--streams 1` gives configs[1]).  Weights: de_streaming_
transformer_xl dimensions, seeded synthetic (no checkpoints offline); audio:
seeded Gaussian noise, already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 6
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...

Prints ONE JSON line (rank 0).  value = whole-job audio-seconds processed per
wall-clock second = number of concurrent real-time streams the job sustains.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from speechcatcher_amd import synth  # noqa: E402
from speechcatcher_amd.config import XL, SearchConfig  # noqa: E402

CHUNK = 10240
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # same guide: HBM3E ~8 TB/s


def build_batch(n_streams, beam, bbd, n_steps_total, device):
    from speechcatcher_amd.engine import StreamBatch
    from speechcatcher_amd.hip_backend import HipBackend
    from speechcatcher_amd.weights import PackedWeights
    sd = synth.make_state_dict(XL, 1234)
    mean, std = synth.stats_to_mean_std(synth.make_stats(XL, kind="meanstd"))
    w = PackedWeights(sd, XL, device, mean, std)
    be = HipBackend(device)
    hops = (n_steps_total + 2) * CHUNK / 10240.0       # encoder hops (16 frames each) in the window
    frames = int(16 * hops) + 64
    tokens = min(2048, int(14 * hops) + 32)
    sb = StreamBatch(w, be, n_streams, SearchConfig(beam_size=beam, use_bbd=bbd),
                     max_frames=frames, max_tokens=tokens,
                     pcm_capacity=CHUNK * (n_steps_total + 2), max_chunk_samples=CHUNK)
    return sb, be


def preload_audio(sb, n_steps_total, stream_offset=0):
    """Synthetic audio for every stream, resident in HBM before timing."""
    n = CHUNK * n_steps_total
    for s in range(sb.S):
        a = synth.synth_audio(stream_offset + s, n)
        sb.pcm[s, :n].copy_(torch.from_numpy(a))
    torch.cuda.synchronize()


OVERLAP = False   # --overlap: prefetch the next step's encoder stage on a second HIP stream


def run_steps(sb, n):
    """n chunk steps over all streams.  With OVERLAP the frontend + encoder pass of
    step i+1 is launched on a second HIP stream before the decode loop of step i
    (StreamBatch.push(prefetch=...)): steady-state pipelining of the same work."""
    items = [(s, CHUNK, False) for s in range(sb.S)]
    for _ in range(n):
        sb.push(items, pcm_resident=True, prefetch=items if OVERLAP else None)


def cpu_baseline(budget_s=20.0, beam=10, bbd=False, warm_calls=4, max_steps=40):
    """The oracle (port of the reference path) timed on this box's host cores."""
    from oracle.ref_port import RefPortModel, RefPortStreaming
    from speechcatcher_amd.mel import melscale_fbanks_slaney
    # 8 intra-op threads: the best single-process setting measured for the
    # reference; torch's default of one thread per host
    # core (128 on the GPU box) is slower on these small ops.
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = synth.make_state_dict(XL, 1234)
    mean, std = synth.stats_to_mean_std(synth.make_stats(XL, kind="meanstd"))
    mel = melscale_fbanks_slaney(257, 0.0, 8000.0, 80, 16000)
    model = RefPortModel(sd, XL, mel, mean, std)
    s = RefPortStreaming(model, beam_size=beam, ctc_weight=0.3, use_bbd=bbd)
    audio = synth.synth_audio(0, CHUNK * (warm_calls + max_steps))
    for i in range(warm_calls):
        s(audio[i * CHUNK:(i + 1) * CHUNK], is_final=False)
    t0 = time.perf_counter()
    n = 0
    while n < max_steps and time.perf_counter() - t0 < budget_s:
        i = warm_calls + n
        s(audio[i * CHUNK:(i + 1) * CHUNK], is_final=False)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * CHUNK / 16000.0 / dt, 4), "unit": "audio_s/s",
            "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"1 stream, chunk steps {warm_calls}..{warm_calls + n - 1} of stream 0 "
                      f"({n} steps, {dt:.1f} s wall), beam {beam}, bbd {int(bbd)}, XL dims, torch-CPU oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--streams", type=int, default=128, help="streams per GPU")
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--bbd", type=int, default=0, help="block boundary detection (reference CLI default: on)")
    ap.add_argument("--chunk", type=int, default=10240,
                    help="samples per chunk step (10240 = 640 ms = one encoder hop; also 8192 = CLI default, 25600 = block size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=2,
                    help="extra (untimed for `value`) steps with per-launch HIP-event timing of the GEMM kernel")
    ap.add_argument("--defer", type=int, default=-1,
                    help="deferred stragglers: end a chunk step's decode loop when at most this many streams are "
                         "still inside their block; they resume in the next step's loop (StreamBatch."
                         "set_defer_threshold; identical per-stream results, a stream is never more than one block "
                         "behind).  Pending blocks are flushed inside the timed region.  -1 (default): 3/8 of the "
                         "streams (measured optimum); 0: strict lock-step, every block completes inside its chunk step.")
    ap.add_argument("--defer-lag", type=int, default=1,
                    help="blocks (chunk periods) a deferred stream may be behind (1: results at most one chunk "
                         "period late; 2 measured +10 %% throughput with threshold 80)")
    ap.add_argument("--overlap", action="store_true",
                    help="launch the frontend + encoder of step i+1 on a second HIP stream before the decode loop of "
                         "step i (measured slower than the serial order: DESIGN.md section 4, negative results)")
    args = ap.parse_args()
    global CHUNK, OVERLAP
    CHUNK = args.chunk
    OVERLAP = bool(args.overlap)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  SC_DIST_BACKEND=gloo + SC_BENCH_SINGLE_DEVICE=1 lets the
        # multi-rank control flow be smoke-tested on a 1-GPU box (collectives on CPU tensors).
        backend = os.environ.get("SC_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get("SC_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    device = f"cuda:{local_rank}"
    coll_device = device if (dist is None or dist.get_backend() == "nccl") else "cpu"
    torch.cuda.set_device(device)

    total_steps = args.warmup + args.steps + args.roofline_steps
    sb, be = build_batch(args.streams, args.beam, bool(args.bbd), total_steps, device)
    preload_audio(sb, total_steps, stream_offset=rank * args.streams)

    if args.defer < 0:
        args.defer = (3 * args.streams) // 8
    sb.set_defer_threshold(args.defer, args.defer_lag)
    run_steps(sb, args.warmup)
    sb.flush()
    torch.cuda.synchronize()
    if sb.timing is not None:
        sb.timing.clear()
    if dist is not None:
        dist.barrier()
    steps0 = sum(st.n_steps_total for st in sb.st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(sb, args.steps)
    sb.flush()                      # deferred blocks of the last steps belong to the timed work
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dec_steps = (sum(st.n_steps_total for st in sb.st) - steps0) / float(args.streams)
    # Roofline leg: the SAME workload continues for a few more steps with hipGraph
    # replay switched off, so that every GEMM launch can be bracketed by HIP
    # events on its launch stream (kernels inside a graph replay cannot be).
    NK = 9   # scasr.h: SC_PROF_KINDS
    ms = (C.c_double * NK)()
    fl = (C.c_double * NK)()
    nn = (C.c_longlong * NK)()
    by = (C.c_double * NK)()
    ev_over_ms = 0.0
    xattn_bytes = 0.0
    if args.roofline_steps > 0:
        be.use_graphs = False
        be.lib.sc_prof_enable(1)
        sb.stats["xattn_rows"] = 0
        run_steps(sb, args.roofline_steps)
        torch.cuda.synchronize()
        be.lib.sc_prof_enable(0)
        be.lib.sc_prof_collect_kinds(ms, fl, by, nn, NK)
        be.use_graphs = True
        ev_over_ms = float(be.lib.sc_prof_event_overhead_ms(sb.stream.cuda_stream))
        # cross-attention: K|V rows of every active stream are read once per layer and step
        xattn_bytes = float(sb.stats.get("xattn_rows", 0)) * 2 * sb.cfg.d_model * 4

    if dist is not None:
        t = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the path's single collective: gather of final token ids
        from speechcatcher_amd.distributed import gather_final_hypotheses, pack_hypotheses
        hy = [sb.hypotheses(s) for s in range(args.streams)]
        ids, sc = pack_hypotheses([h[0]["yseq"] if h else [] for h in hy],
                                  [h[0]["score"] if h else 0.0 for h in hy], 256, coll_device)
        gathered = gather_final_hypotheses(ids, sc, args.streams)
        assert len(gathered) == world

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    audio_s = world * args.streams * args.steps * CHUNK / 16000.0
    value = audio_s / elapsed
    # dominant kernel of the path = the kernel kind with the largest summed launch time
    # in the roofline leg (agrees with the rocprofv3 summary under profiles/)
    names = ["gemm_naive_kernel", "gemm_skinny_kernel", "gemm_mfma_kernel<128,128>", "gemm_mfma_kernel<64,64>",
             "proj_ln_proj_kernel<256,*>", "ffn_fused_kernel<256,*>", "dec_attn_flash_kernel<32,10,self>",
             "dec_attn_flash_kernel<32,10,cross>", "rowtile_proj_kernel<256,*>"]
    net = [max(ms[i] - nn[i] * ev_over_ms, 0.0) for i in range(NK)]
    tot_ms = max(sum(net), 1e-9)
    per_kernel = []
    for i in range(NK):
        if nn[i] == 0:
            continue
        ent = {"kernel": names[i], "launches": int(nn[i]), "avg_launch_us": round(net[i] * 1e3 / nn[i], 2),
               "share_of_timed_kernel_time": round(net[i] / tot_ms, 4)}
        if fl[i] > 0:
            ent["tflops"] = round(fl[i] / (max(net[i], 1e-9) * 1e-3) / 1e12, 3)
            ent["frac_of_f32_mfma_peak"] = round(ent["tflops"] / PEAK_F32_MFMA_TFLOPS, 4)
        if i == 7 and xattn_bytes > 0:
            ent["hbm_gbs_algorithmic"] = round(xattn_bytes / (max(net[i], 1e-9) * 1e-3) / 1e9, 1)
            ent["frac_of_hbm_peak"] = round(ent["hbm_gbs_algorithmic"] / PEAK_HBM_GBS, 4)
        per_kernel.append(ent)
    roof = None
    if any(nn[i] > 0 for i in range(NK)):
        v = max(range(NK), key=lambda i: net[i])
        raw_us = ms[v] * 1e3 / nn[v]
        t_ms = max(net[v], 1e-9)
        common = {"kernel": names[v], "traffic": None,
                  "avg_launch_us": round(t_ms * 1e3 / nn[v], 2), "avg_launch_us_raw_events": round(raw_us, 2),
                  "event_pair_overhead_us": round(ev_over_ms * 1e3, 2), "launches_timed": int(nn[v]),
                  "share_of_timed_kernel_time": round(net[v] / tot_ms, 4),
                  "measured_over": f"{args.roofline_steps} steps following the timed region, same workload, "
                                   "hipGraph replay off, HIP events around every launch",
                  "per_kernel": per_kernel}
        if fl[v] > 0:
            ach = fl[v] / (t_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    "flops_per_launch_avg": round(fl[v] / nn[v] / 1e6, 1),
                    "flops_unit": "MFLOP algorithmic per launch (DESIGN.md section 5)",
                    "algorithmic_bytes_per_launch_avg": int(by[v] / nn[v])}
        else:
            bytes_v = xattn_bytes if v == 7 else 0.0
            ach = bytes_v / (t_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4),
                    "algorithmic_bytes_per_launch_avg": int(bytes_v / nn[v])}
        roof.update(common)

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of
    # this same command (FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE), committed
    # under profiles/; it cannot be collected from inside the process.
    if roof is not None and args.streams == 128 and not args.bbd:
        tpath = os.path.join(ROOT, "profiles", "r01_bench_default_pmc_hbm_traffic.csv")
        key = roof["kernel"].split("<")[0]
        if os.path.exists(tpath):
            for line in open(tpath).read().splitlines()[1:]:
                cols = line.split(",")
                if key in cols[0] and (("Lb1E" in cols[0]) == ("self" in roof["kernel"]) or "attn" not in key):
                    roof["traffic"] = int(cols[-1])
                    roof["traffic_unit"] = ("HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE*2 + WRITE_SIZE, "
                                            "profiles/r01_bench_default_pmc_hbm_traffic.csv)")
                    break

    single = None
    if not args.no_single_stream and world == 1:
        OVERLAP = False   # latency of ONE real-time stream: the next chunk does not exist yet
        sb1, _ = build_batch(1, args.beam, bool(args.bbd), args.warmup + args.steps, device)
        preload_audio(sb1, args.warmup + args.steps)
        run_steps(sb1, args.warmup)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(sb1, args.steps)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        hop_s = CHUNK / 16000.0
        single = {"ms_per_hop": round(e1 / args.steps * 1e3, 3), "rtf": round(e1 / (args.steps * hop_s), 5),
                  "x_realtime": round(args.steps * hop_s / e1, 1)}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(beam=args.beam, bbd=bool(args.bbd))
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}

    out = {
        "metric": f"concurrent real-time streams (audio-seconds/s), de_xl dims, {CHUNK * 1000 // 16000} ms ({CHUNK}-sample) chunk steps, beam 10 CTC+attention",
        "value": round(value, 2), "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"de_streaming_transformer_xl dims, {args.streams} concurrent synthetic streams/GPU "
                               f"(batched encoder + batched beam), beam {args.beam}, chunk 10240 samples, bbd {args.bbd}",
                   "streams_per_gpu": args.streams, "chunk_samples": CHUNK, "beam": args.beam, "bbd": args.bbd,
                   "deferred_stragglers": args.defer, "deferred_max_lag_blocks": args.defer_lag,
                   "pipelining": ("encoder of chunk step i+1 on a second HIP stream overlaps the decode loop of step i"
                                  if args.overlap else "none"),
                   "parallelism": f"streams sharded x{world}, no collective in the hot loop"},
        "chunk_steps_per_s": round(world * args.streams * args.steps / elapsed, 2),
        "decode_steps_per_hop": round(dec_steps / max(args.steps, 1), 2),
        "roofline": roof, "cpu_baseline": cpu, "single_stream": single,
    }
    if sb.timing is not None:
        out["host_phase_ms_per_step"] = {k: round(v / args.steps * 1e3, 3) for k, v in sb.timing.items()}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
