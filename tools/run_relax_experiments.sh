#!/bin/bash
# round-2 relax experiments: what-if timings, then PMC passes (FETCH_SIZE; L2 hit/miss + WRITE_SIZE) per configuration
set -o pipefail
O=gpurun_out/r02_relax
mkdir -p $O
python3 tools/whatif.py 512 0 > $O/whatif_ilv0.log 2>&1 && python3 tools/whatif.py 512 1 > $O/whatif_ilv1.log 2>&1 || exit 1
cat $O/whatif_ilv0.log $O/whatif_ilv1.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "1 1" "1 2" "1 3" "1 4"; do
  set -- $cfg
  rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_fetch_ilv$1_wi$2 -o out --output-format csv -- python3 $R/tools/relax_solver.py 512 4 $1 $2 > $R/$O/pmc_fetch_ilv$1_wi$2.log 2>&1 || exit 1
  python3 $R/tools/pmc_sum.py $R/$O/pmc_fetch_ilv$1_wi$2 > $R/$O/pmc_fetch_ilv$1_wi$2.txt
  echo "== fetch ilv $1 whatif $2"; head -4 $R/$O/pmc_fetch_ilv$1_wi$2.txt
done
for cfg in "0 0" "1 0"; do
  set -- $cfg
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE -d $R/$O/pmc_l2_ilv$1 -o out --output-format csv -- python3 $R/tools/relax_solver.py 512 4 $1 0 > $R/$O/pmc_l2_ilv$1.log 2>&1 || exit 1
  python3 $R/tools/pmc_sum.py $R/$O/pmc_l2_ilv$1 > $R/$O/pmc_l2_ilv$1.txt
  echo "== l2 ilv $1"; head -8 $R/$O/pmc_l2_ilv$1.txt
done
# keep the merge small: drop the raw counter csv files
find $R/$O -name "*.csv" -size +2M -delete
