#!/bin/bash
# end-of-round validation on the GPU box: full GPU test suite, the four bench workloads, rocprofv3 kernel stats of the
# default bench command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE) of the level-0 relax launch.
# usage: bash tools/final_validation.sh TAG      (writes gpurun_out/TAG/)
set -o pipefail
TAG=${1:-r02_final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py > $O/bench_3d27.json 2> $O/bench_3d27.err; echo "bench 3d27 rc=$?"
for wl in 2d9 2d9l 2d5; do python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/rocprof_3d27 -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --allocations 1 > $O/bench_3d27_under_rocprof.json 2> $O/rocprof_3d27.err; echo "rocprof stats rc=$?"
cp $O/rocprof_3d27/out_kernel_stats.csv $O/rocprof_kernel_stats_3d27.csv 2>/dev/null || find $O/rocprof_3d27 -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_3d27.csv \;
rocprofv3 --kernel-trace --stats -d $O/rocprof_2d9l -o out --output-format csv -- python3 $R/bench.py --workload 2d9l --no-cpu-baseline > $O/bench_2d9l_under_rocprof.json 2> $O/rocprof_2d9l.err
find $O/rocprof_2d9l -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_2d9l.csv \;
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o out --output-format csv -- python3 $R/tools/relax_solver.py 512 4 320 0 > $O/pmc_fetch.log 2>&1; python3 $R/tools/pmc_sum.py $O/pmc_fetch > $O/pmc_fetch_size_relax512.txt
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o out --output-format csv -- python3 $R/tools/relax_solver.py 512 4 320 0 > $O/pmc_write.log 2>&1; python3 $R/tools/pmc_sum.py $O/pmc_write > $O/pmc_write_size_relax512.txt
head -6 $O/pmc_fetch_size_relax512.txt $O/pmc_write_size_relax512.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cat $O/bench_3d27.json
