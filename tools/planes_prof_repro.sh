R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_planes_repro; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $O/run$i -o p --output-format csv -- python3 $R/tools/planes_prof_repro.py $O/maps$i.txt 64 8 > $O/run$i.log 2>&1
  echo "run $i rc=$?"; grep -c SIGSEGV $O/run$i.log
  if grep -q SIGSEGV $O/run$i.log; then break; else rm -f $O/maps$i.txt; fi
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
