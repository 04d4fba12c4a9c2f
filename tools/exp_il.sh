#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "ffn" 2>&1 | tail -3
python -m pytest tests/test_gpu_baseline_size.py -x -q -k "stream_resident or bit_reproducible" 2>&1 | tail -3
REPS=3 bash tools/ab_libs.sh "--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --roofline-steps 0" product build_ab/libscasr_noil.so 2>&1 | tee gpurun_out/r06_ab_ffn_interleave.txt
