#!/bin/bash
# kernel-trace statistics of the split-bf16 training step to gpurun_out/x3_now.md
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_x3
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/prof_x3 -- python $R/bench.py --precision 32 --steps 6 --warmup 2 --regions 1 --no-cpu-baseline --no-sub-benches --no-roofline-pass > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_x3 -name "*.db" | head -1) > $R/gpurun_out/x3_now.md
rm -rf $R/gpurun_out/prof_x3
grep -E "attn|split_rows" $R/gpurun_out/x3_now.md | cut -c1-150
tail -1 $R/gpurun_out/x3_now.md
