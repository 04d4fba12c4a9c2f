#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_s256_ffn.txt \
  "s256_default||--streams 256 --steps 10" \
  "s256_ffn_5_2|SC_DEC_FFN_FORCE=2000,5,2|--streams 256 --steps 10" \
  "s256_ffn_5_1|SC_DEC_FFN_FORCE=2000,5,1|--streams 256 --steps 10" \
  "s256_ffn_3_2|SC_DEC_FFN_FORCE=2000,3,2|--streams 256 --steps 10" \
  "s256_stream_min_1921|SC_STREAM_MIN=1921|--streams 256 --steps 10" \
  "s256_stream_min_961|SC_STREAM_MIN=961|--streams 256 --steps 10"
