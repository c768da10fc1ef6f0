for s in ${STAG:-0 3}; do for rep in 1 2 3; do echo "stagger $s"; LK_WINP_STAGGER=$s timeout 200 python tools/winp_bench.py $1 2>&1 | grep -v amdgpu.ids | grep "persistent"; done; done
