set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_shared_iterate.py -x -q -m gpu > gpurun_out/shared_test.log 2>&1 || { tail -40 gpurun_out/shared_test.log; exit 1; }
tail -3 gpurun_out/shared_test.log
export QPN_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 50 --warmup 10 > gpurun_out/b4_p2p.log 2>&1 || { tail -30 gpurun_out/b4_p2p.log; exit 1; }
tail -1 gpurun_out/b4_p2p.log | cut -c1-200; tail -1 gpurun_out/b4_p2p.log | grep -o '"exchange": "[^"]*"'
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 50 --warmup 10 --scaling strong > gpurun_out/b4_p2p_s.log 2>&1 || { tail -30 gpurun_out/b4_p2p_s.log; exit 1; }
tail -1 gpurun_out/b4_p2p_s.log | cut -c1-200; tail -1 gpurun_out/b4_p2p_s.log | grep -o '"exchange": "[^"]*"'
