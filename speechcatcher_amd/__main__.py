"""``python -m speechcatcher_amd <media.wav>`` - the file mode of the reference CLI on the MI355X engine.

Flags follow speechcatcher/speechcatcher.py:756-808 (``main``) and the file path of ``recognize_file`` /
``recognize`` (:358-400, :414-572): 16 kHz mono recording -> endpointing into segments (> 60 s) -> chunked
streaming decode of every segment (``--chunk-length`` samples per call, is_final on the last chunk of a segment,
reset() after it) -> paragraphs -> ``<input>.txt`` and ``<input>.json``.

Differences to the reference CLI, all on the host side: only ``--decoder native`` exists here; the device is a ROCm
GPU (no CPU path); ``-n`` is the number of parallel stream slots of ONE model replica instead of worker processes
(default 1 = the reference's behaviour on a GPU: segments decoded serially on one model, speechcatcher.py:823-825);
input must already be 16 kHz mono 16-bit WAV (the ffmpeg conversion, microphone mode and model download are out of
scope - pass a local model directory or a .scasr blob as ``-m``).
"""
import argparse
import json
import logging
import os
import sys
import wave

import numpy as np


def read_wav_16k_mono(path):
    with wave.open(path, "rb") as f:
        ch, bits, rate = f.getnchannels(), f.getsampwidth(), f.getframerate()
        buf = f.readframes(-1)
    if ch != 1 or bits != 2 or rate != 16000:
        raise SystemExit(f"Error: '{path}' is {ch} channel(s), {8 * bits} bit, {rate} Hz; this entry point reads 16 kHz mono "
                         "16-bit WAV only (the reference converts other media with ffmpeg first: "
                         "ffmpeg -i in -ac 1 -ar 16000 out.wav)")
    return np.frombuffer(buf, dtype="<i2"), rate


def recognize_file(speech2text, media_path, output_file="", quiet=True, progress=True, num_slots=1, chunk_length=8192):
    """speechcatcher.py:358-400: recording -> text + paragraphs JSON next to the input."""
    from .config import SearchConfig
    from .native import NativeStreamBatch
    from .segmenter import recognize_recording
    raw, rate = read_wav_16k_mono(media_path)
    seconds = len(raw) / float(rate)
    # capacities for the longest segment the endpointer may produce (it cuts at <= 180 s: simple_endpointing)
    seg_s = min(seconds, 200.0) + 2.0
    batch = NativeStreamBatch(speech2text.weights, max(1, num_slots),
                              SearchConfig(beam_size=speech2text.beam_size, ctc_weight=speech2text.ctc_weight,
                                           use_bbd=speech2text.use_bbd),
                              max_frames=int(seg_s * 25) + 64, max_tokens=min(2048, int(seg_s * 12) + 64),
                              pcm_capacity=max(1 << 20, int(seg_s * rate) + chunk_length),
                              max_chunk_samples=max(chunk_length, 32768), engine=speech2text.batch.engine)
    # finalize_all only with the very last chunk of the recording, like the reference CLI (speechcatcher.py:586)
    text, info = recognize_recording(batch, raw, rate, chunk_length=chunk_length, token_list=speech2text.token_list,
                                     reference_finalize=True)
    out_txt, out_json = (output_file or media_path) + ".txt", (output_file or media_path) + ".json"
    with open(out_txt, "w") as f:
        f.write(text)
    complete = {"complete_text": text, "paragraphs": info}
    with open(out_json, "w") as f:
        f.write(json.dumps(complete, indent=4))
    if not quiet:
        sys.stdout.write(text)
    print(f"Wrote transcription to {out_txt} and {out_json}.")
    return complete


def main(argv=None):
    p = argparse.ArgumentParser(prog="python -m speechcatcher_amd",
                                description="Decode speech with speechcatcher models on an MI355X (native decoder path).")
    p.add_argument("-l", "--live-transcription", dest="live", action="store_true",
                   help="(not available here: microphone mode is host I/O of the reference CLI)")
    p.add_argument("-m", "--model", dest="model", default="de_streaming_transformer_xl",
                   help="model tag (needs espnet_model_zoo, like the reference), a local model directory, or a .scasr blob")
    p.add_argument("-d", "--device", dest="device", default="cuda", help="'cuda' (there is no CPU path)")
    p.add_argument("-b", "--beamsize", dest="beamsize", type=int, default=5, help="beam size for the decoder")
    p.add_argument("--decoder", dest="decoder", choices=["native", "espnet"], default="native",
                   help="only 'native' is implemented")
    p.add_argument("--fp16", dest="fp16", action="store_true",
                   help="accepted like the reference does: the native decoder stays fp32 (speechcatcher.py:205-210)")
    p.add_argument("--disable-bbd", dest="disable_bbd", action="store_true", help="disable block boundary detection")
    p.add_argument("--quiet", dest="quiet", action="store_true")
    p.add_argument("--no-progress", dest="no_progress", action="store_true")
    p.add_argument("--num-threads", dest="num_threads", type=int, default=1, help="(host threads: unused on the GPU path)")
    p.add_argument("--cache-dir", dest="cache_dir", default="~/.cache/espnet")
    p.add_argument("-n", "--num-processes", dest="num_processes", type=int, default=-1,
                   help="parallel stream slots for the segments of a long recording (-1: 1, the reference's GPU behaviour)")
    p.add_argument("--chunk-length", dest="chunk_length", type=int, default=8192,
                   help="raw audio samples per streaming call (default 8192)")
    p.add_argument("--log-level", dest="log_level", default="ERROR", choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"])
    p.add_argument("--show-ffmpeg-output", dest="show_ffmpeg_output", action="store_true", help="(unused: no ffmpeg step)")
    p.add_argument("inputfile", nargs="?", default="", help="input recording (16 kHz mono 16-bit WAV)")
    args = p.parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level))
    if args.decoder != "native":
        raise SystemExit("Error: only --decoder native is implemented by speechcatcher_amd")
    if args.live:
        raise SystemExit("Error: live transcription (microphone) is not part of this package")
    if args.inputfile == "":
        p.print_help()
        return 0
    if not os.path.isfile(args.inputfile):
        print(f"Error: Input file '{args.inputfile}' does not exist or is not a valid file.")
        return -1
    from .speech2text_streaming import load_model
    speech2text = load_model(tag=args.model, device=args.device, beam_size=args.beamsize, quiet=args.quiet,
                             cache_dir=args.cache_dir, decoder_impl=args.decoder, fp16=args.fp16,
                             use_bbd=not args.disable_bbd)
    recognize_file(speech2text, args.inputfile, quiet=args.quiet or not args.no_progress, progress=not args.no_progress,
                   num_slots=1 if args.num_processes < 1 else args.num_processes, chunk_length=args.chunk_length)
    return 0


if __name__ == "__main__":
    sys.exit(main())
