#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_overlap64_hpw4.txt \
  "s64_hpw4|SC_DEC_HPW=4|--streams 64" \
  "s64_hpw4_enc128cu|SC_DEC_HPW=4 SC_ENC_CUS=128|--streams 64" \
  "s64_hpw4_enc112cu|SC_DEC_HPW=4 SC_ENC_CUS=112|--streams 64" \
  "s64_hpw4_enc144cu|SC_DEC_HPW=4 SC_ENC_CUS=144|--streams 64" \
  "s64_hpw4_enc64cu|SC_DEC_HPW=4 SC_ENC_CUS=64|--streams 64" \
  "s64_hpw1||--streams 64" \
  "s64_hpw4_serial|SC_DEC_HPW=4 SC_ENC_OVERLAP=0|--streams 64" \
  "s128_serial|SC_ENC_OVERLAP=0|"
export SC_TEST_HOOKS=1
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0 --streams 64"
export SC_DEC_HPW=4
for cu in 0 128; do
  export SC_ENC_CUS=$cu
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/r06_s64h4_cu${cu}_prof.log 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1)
  python tools/rocpd_phases.py $DB 150 > gpurun_out/r06_s64h4_cu${cu}_phases.txt 2>&1
  python tools/rocpd_stats.py $DB gpurun_out/r06_s64h4_cu${cu}_kernel_stats.csv > /dev/null
done
