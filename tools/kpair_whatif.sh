#!/bin/bash
# round 2: would a k-pair pass (a run of rows of plane k, then the same rows of plane k-1, in one workgroup) turn the
# double read of the inter-plane slot-rows into cache hits?  Timing by run length, then FETCH_SIZE of the fused launch
# (CEDAR_AMD_WHATIF=8: dependencies between workgroups ignored, results wrong, traffic and timing representative).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_kpair; mkdir -p $O
cd $R
for fr in 8 4 2 1; do
  echo "== CEDAR_AMD_FRUN=$fr"
  CEDAR_AMD_FRUN=$fr python3 tools/kpair_time.py || exit 1
done 2>&1 | tee $O/kpair_timing.log
cd /tmp && export TMPDIR=/tmp
for fr in 8 2; do
  export CEDAR_AMD_FRUN=$fr
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_frun$fr -o out --output-format csv -- python3 $R/tools/relax_solver.py 512 4 320 8 > $O/pmc_fetch_frun$fr.log 2>&1
  python3 $R/tools/pmc_sum.py $O/pmc_fetch_frun$fr | grep relax27 | tee $O/pmc_fetch_frun$fr.txt
done
find $O -name "*.csv" -size +1M -delete
