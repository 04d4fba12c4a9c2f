#!/bin/bash
# Profiles of the default bench command for profiles/ (run on the GPU box: gpurun -- 'bash tools/prof_bench.sh r03'):
#   1. the bench line itself                                  -> gpurun_out/<tag>_bench_default.json
#   2. rocprofv3 --kernel-trace of the same command           -> <tag>_bench_default_kernel_stats.csv (per-kernel calls / avg)
#   3. separate --pmc FETCH_SIZE / WRITE_SIZE passes          -> <tag>_bench_default_pmc_hbm_traffic.csv (HBM bytes per launch;
#      (never combined with a trace domain other than kernel-trace)      FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md)
#   4. a --pmc pass of the matrix-core / LDS counters         -> <tag>_bench_default_pmc_sq.csv
# The profiled runs time the headline leg only (same window, same steps); the extra legs are switched off.
# Optional: a name and extra bench arguments for a non-default mode, e.g.
#   bash tools/prof_bench.sh r03 split16 "--ffn-dtype split16"   -> gpurun_out/r03_bench_split16*
TAG=${1:-r06}
NAME=${2:-default}
EXTRA=${3:-}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py $EXTRA > gpurun_out/${TAG}_bench_${NAME}.json 2> gpurun_out/${TAG}_bench_${NAME}.err
ARGS="$EXTRA --legs none"
# (round 5) the roofline leg - same workload, graph replay off - is bracketed by sc_marker kernels: every pass is cut to the launches
# between them, so that durations, counters and bench.py's algorithmic bytes belong to the SAME launches
rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS --roofline-csv gpurun_out/${TAG}_bench_${NAME}_roofline_leg.csv > gpurun_out/${TAG}_${NAME}_prof_kernel.log 2>&1
python tools/rocpd_stats.py $(find /tmp/pk -name "*.db" | head -1) gpurun_out/${TAG}_bench_${NAME}_kernel_stats.csv > /dev/null
rm -rf /tmp/pf; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/pf -- python3 bench.py $ARGS > gpurun_out/${TAG}_${NAME}_prof_fetch.log 2>&1
rm -rf /tmp/pw; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/pw -- python3 bench.py $ARGS > gpurun_out/${TAG}_${NAME}_prof_write.log 2>&1
python tools/pmc_traffic_csv.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) gpurun_out/${TAG}_bench_${NAME}_pmc_hbm_traffic.csv > /dev/null 2>&1 || echo "pmc join failed"
python tools/roofline_window.py gpurun_out/${TAG}_bench_${NAME}_roofline_leg.csv $(find /tmp/pk -name "*.db" | head -1) $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) gpurun_out/${TAG}_bench_${NAME}_roofline_window.csv > /dev/null || echo "roofline window join failed"
rm -rf /tmp/ps; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format rocpd -d /tmp/ps -- python3 bench.py $ARGS --steps 6 --roofline-steps 0 > gpurun_out/${TAG}_${NAME}_prof_sq.log 2>&1
python tools/pmc_sq_csv.py $(find /tmp/ps -name "*.db" | head -1) gpurun_out/${TAG}_bench_${NAME}_pmc_sq.csv > /dev/null 2>&1 || echo "sq pmc summary failed"
tail -c 600 gpurun_out/${TAG}_${NAME}_prof_sq.log
