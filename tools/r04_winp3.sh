echo "full: $(timeout 200 python tools/winp_bench.py 2>&1 | grep persistent | cut -c1-75)"
for a in 8 10; do echo "ablate $a: $(LK_LIB=$GRAFT_REPO_ROOT/laplace_amd/csrc/liblaplace_hip_a$a.so timeout 200 python tools/winp_bench.py 2>&1 | grep persistent | cut -c1-75)"; done
