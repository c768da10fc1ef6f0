for rep in 1 2; do
echo "A (default lib)"; timeout 200 python tools/winp_bench.py --all 2>&1 | grep "persistent"
echo "B (variant lib)"; LK_LIB=$GRAFT_REPO_ROOT/laplace_amd/csrc/liblaplace_hip_b.so timeout 200 python tools/winp_bench.py --all 2>&1 | grep "persistent"
done
