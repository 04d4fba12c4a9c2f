#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_ffnforce.txt \
  "default||" \
  "ffn_5_1|SC_DEC_FFN_FORCE=1000,5,1|" \
  "ffn_4_1|SC_DEC_FFN_FORCE=1000,4,1|" \
  "ffn_2_2|SC_DEC_FFN_FORCE=1000,2,2|" \
  "default2||"
