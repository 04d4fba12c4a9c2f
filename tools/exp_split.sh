#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_baseline_size.py -x -q -k "stream_resident or bit_reproducible" 2>&1 | tail -3
bash tools/ab_env.sh gpurun_out/r06_ab_split_ffn.txt \
  "split||" \
  "pro|SC_DEC_FFN_SPLIT=0|" \
  "split_5_1|SC_DEC_FFN_FORCE=1000,5,1|" \
  "split_3_2|SC_DEC_FFN_FORCE=1000,3,2|" \
  "split_2||" \
  "pro_2|SC_DEC_FFN_SPLIT=0|"
