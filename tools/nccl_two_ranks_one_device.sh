cd $GRAFT_REPO_ROOT
PORT=29611
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT SC_DIST_BACKEND=nccl SC_BENCH_SINGLE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout 240 python bench.py --gpus 2 --streams 8 --steps 3 --warmup 2 --preroll 3 --roofline-steps 0 --no-cpu-baseline --no-long-context > gpurun_out/nccl_rank$r.out 2> gpurun_out/nccl_rank$r.err &
done
wait
tail -c 300 gpurun_out/nccl_rank0.out; echo; grep -i "error\|nccl\|rccl" gpurun_out/nccl_rank0.err | tail -5; grep -i "error\|nccl" gpurun_out/nccl_rank1.err | tail -3
