#!/bin/bash
# round-3 measurement pass on the GPU box: default bench line, rocprofv3 kernel stats of the same command, and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE, separate runs) over level-0 sweeps of the partial-sum relax.
# usage: bash tools/r03_measure.sh TAG      (writes gpurun_out/TAG/)
set -o pipefail
TAG=${1:-r03_measure}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench_3d27.json 2> $O/bench_3d27.err; echo "bench 3d27 rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/rocprof_3d27 -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --allocations 1 --no-other-workloads > $O/bench_3d27_under_rocprof.json 2> $O/rocprof_3d27.err; echo "rocprof stats rc=$?"
find $O/rocprof_3d27 -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_3d27.csv \;
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o out --output-format csv -- python3 $R/tools/psum_relax.py 512 4 > $O/pmc_fetch.log 2>&1; python3 $R/tools/pmc_sum.py $O/pmc_fetch > $O/pmc_fetch_size_psum512.txt
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o out --output-format csv -- python3 $R/tools/psum_relax.py 512 4 > $O/pmc_write.log 2>&1; python3 $R/tools/pmc_sum.py $O/pmc_write > $O/pmc_write_size_psum512.txt
grep relax27 $O/pmc_fetch_size_psum512.txt $O/pmc_write_size_psum512.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cat $O/bench_3d27.json
