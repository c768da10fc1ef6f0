mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/fit_ab.py 20 5 early=early_flush:True late=early_flush:False 2>&1 | tail -3
