#!/bin/bash
# Where does a 2-rank bench on one GPU hang?  Runs it in the lab build with CC_GEMM_Q4 = 1 / 0, and after 100 s sends SIGABRT to the rank
# processes (PYTHONFAULTHANDLER=1 dumps every thread's Python stack).  usage (GPU box): bash tools/diag/hang_probe.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
export CC_BENCH_DEVICE=0 CC_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 PYTHONFAULTHANDLER=1 CLIPCAP_HIP_LIB=lab
for q in 1 0; do
    CC_GEMM_Q4=$q python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$q $R/bench.py --gpus 2 --steps 2 --warmup 1 \
        --no-cpu-baseline --batch 64 > $R/gpurun_out/mr_q$q.log 2>&1 &
    TP=$!
    for i in $(seq 1 100); do sleep 1; kill -0 $TP 2>/dev/null || break; done
    if kill -0 $TP 2>/dev/null; then
        echo "CC_GEMM_Q4=$q: still running after 100 s -> dumping stacks"
        for c in $(pgrep -P $TP); do kill -ABRT $c; done
        sleep 5
        kill $TP 2>/dev/null
        wait $TP 2>/dev/null
    else
        wait $TP; echo "CC_GEMM_Q4=$q: finished rc=$?"
    fi
    grep -v 'Gloo\|socket.cpp\|OMP_NUM\|\*\*\*\*' $R/gpurun_out/mr_q$q.log | tail -60
done
