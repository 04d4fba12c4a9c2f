#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "ffn" 2>&1 | tail -2
python -m pytest tests/test_gpu_baseline_size.py -x -q -k "stream_resident or bit_reproducible" 2>&1 | tail -2
bash tools/ab_env.sh gpurun_out/r06_ab_ffn_xcd.txt \
  "xcd_map||" \
  "old_map|SC_FFN_XCD=0|" \
  "xcd_map_2||" \
  "old_map_2|SC_FFN_XCD=0|" \
  "xcd_map_3||" \
  "old_map_3|SC_FFN_XCD=0|" \
  "s256_xcd||--streams 256 --steps 10" \
  "s256_old|SC_FFN_XCD=0|--streams 256 --steps 10"
