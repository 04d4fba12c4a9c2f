#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "lockstep_xl and stream_resident" 2>&1 | tail -3
python -m pytest tests/test_gpu_baseline_size.py -x -q -k "stream_resident" 2>&1 | tail -3
SC_TEST_HOOKS=1 SC_LIB_VARIANT=build_ab/libscasr_phase.so python tools/stream_phase_times.py 128 36 0 2>&1 | tail -13
bash tools/ab_env.sh gpurun_out/r06_ab_stream5.txt \
  "hpw4|SC_DEC_STREAM=0|" \
  "stream||" \
  "s256_hpw4|SC_DEC_STREAM=0|--streams 256 --steps 10" \
  "s256_stream||--streams 256 --steps 10"
