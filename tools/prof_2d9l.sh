#!/bin/bash
# kernels of one line-xy V-cycle at 8192^2 (BASELINE config 3) under rocprofv3
OUT=$PWD/gpurun_out/prof${WL:-2d9l}_${1:-a}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-2d9l} --steps 3 --warmup 1 --allocations 1 --no-cpu-baseline --no-other-workloads > $OUT/bench.log 2>&1
find $OUT/prof -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
