mkdir -p gpurun_out/r2x
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/bench_comm_overhead.py 2>&1 | tail -2 | tee gpurun_out/r2x/comm_overhead.json
