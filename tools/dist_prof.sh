#!/bin/bash
# kernel breakdown of one rank of a rank grid on the native driver with the loop-back transport
# usage: tools/dist_prof.sh [grid=2x2x2] [tag]
G=${1:-2x2x2}; TAG=${2:-$G}
OUT=$PWD/gpurun_out/distprof_$TAG; mkdir -p $OUT
python tools/dist_overhead.py 512 $G > $OUT/time.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
DIST_OVERHEAD_ONLY=native rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $GRAFT_REPO_ROOT/tools/dist_overhead.py 512 $G > $OUT/prof.log 2>&1
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
