cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
B="python bench.py --no-cpu-baseline --no-deferred --no-single-stream --roofline-steps 0"
for t in 15 7 5 3; do SC_ENC_START=$t $B > gpurun_out/r02/ov_start$t.json 2>> gpurun_out/r02/ov.err; done
SC_ENC_START=10 SC_ENC_CUS=192 $B > gpurun_out/r02/ov_start10_cu192.json 2>> gpurun_out/r02/ov.err
SC_ENC_START=10 SC_ENC_CUS=224 $B > gpurun_out/r02/ov_start10_cu224.json 2>> gpurun_out/r02/ov.err
SC_ENC_START=5 SC_ENC_CUS=224 $B > gpurun_out/r02/ov_start5_cu224.json 2>> gpurun_out/r02/ov.err
