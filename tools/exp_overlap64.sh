#!/bin/bash
# Round 6 experiment: does the encoder overlap the decode chain when the decode kernels leave CUs free?
# 64 streams (decode layer kernels = 128 workgroups of one CU each) with the encoder stream confined to a CU mask.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_env.sh gpurun_out/r06_ab_overlap64.txt \
  "base128||" \
  "s64||--streams 64" \
  "s64_enc128cu|SC_ENC_CUS=128|--streams 64" \
  "s64_enc96cu|SC_ENC_CUS=96|--streams 64" \
  "s64_enc160cu|SC_ENC_CUS=160|--streams 64" \
  "s96_enc128cu|SC_ENC_CUS=128|--streams 96" \
  "s96||--streams 96"
export SC_TEST_HOOKS=1
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0 --streams 64"
for cu in 0 128; do
  export SC_ENC_CUS=$cu
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/r06_s64_cu${cu}_prof.log 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1)
  python tools/rocpd_phases.py $DB 150 > gpurun_out/r06_s64_cu${cu}_phases.txt 2>&1
  python tools/rocpd_stats.py $DB gpurun_out/r06_s64_cu${cu}_kernel_stats.csv > /dev/null
done
head -12 gpurun_out/r06_s64_cu128_phases.txt
