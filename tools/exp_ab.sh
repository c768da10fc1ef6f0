# A/B of two builds of the library on the same box: default vs laplace_amd/csrc/liblaplace_hip_b.so
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_ab.log; : > $O
B=$GRAFT_REPO_ROOT/laplace_amd/csrc/liblaplace_hip_b.so
for rep in 1 2 3; do
  echo "A (default lib): $(timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O
  echo "B (variant lib): $(LK_LIB=$B timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O
done
cat $O
