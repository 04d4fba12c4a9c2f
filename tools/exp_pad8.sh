#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_grid_pad8.txt \
  "pad8|SC_DEC_GRID_PAD8=1|" \
  "nopad||" \
  "pad8_2|SC_DEC_GRID_PAD8=1|" \
  "nopad_2||" \
  "pad8_3|SC_DEC_GRID_PAD8=1|" \
  "nopad_3||"
