# run-length / level sweep of the partial-sum relax (tools/psum_relax.py); usage: bash tools/psum_levels.sh > log
R=${GRAFT_REPO_ROOT:-.}
cd $R
for n in 192 224 256 320 448 512; do
  echo "== $n default"; python3 tools/psum_relax.py $n 10
  echo "== $n reference order (default run length)"; CEDAR_AMD_PSUM=0 python3 tools/psum_relax.py $n 10
  for f in 8 16 32; do echo "== $n psum frun $f"; CEDAR_AMD_FRUN=$f python3 tools/psum_relax.py $n 10; done
done
