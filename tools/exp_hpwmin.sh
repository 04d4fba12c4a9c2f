#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_hpw_min.txt \
  "hpw_min_640||" \
  "hpw_min_320|SC_HPW_MIN=320|" \
  "hpw_min_480|SC_HPW_MIN=480|" \
  "hpw_min_160|SC_HPW_MIN=160|" \
  "hpw_min_800|SC_HPW_MIN=800|" \
  "hpw_min_640_b||" \
  "strict_640||--mode strict" \
  "strict_320|SC_HPW_MIN=320|--mode strict" \
  "strict_160|SC_HPW_MIN=160|--mode strict"
