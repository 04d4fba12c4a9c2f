#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_stream256.txt \
  "s256_hpw4|SC_DEC_STREAM=0|--streams 256 --steps 10" \
  "s256_stream||--streams 256 --steps 10" \
  "s256_stream_enc128|SC_ENC_CUS=128|--streams 256 --steps 10" \
  "s256_hpw4_qd2|SC_DEC_STREAM=0|--streams 256 --steps 10 --queue-depth 2" \
  "s256_stream_qd2||--streams 256 --steps 10 --queue-depth 2" \
  "s128_hpw4_qd2|SC_DEC_STREAM=0|--queue-depth 2" \
  "s128_stream_qd2||--queue-depth 2" \
  "s128_stream_qd2_enc128|SC_ENC_CUS=128|--queue-depth 2" \
  "s192_hpw4|SC_DEC_STREAM=0|--streams 192 --steps 12" \
  "s192_stream||--streams 192 --steps 12"
