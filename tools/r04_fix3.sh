mkdir -p gpurun_out; export TMPDIR=/tmp
LK_SYNC_BEFORE_FINALIZE=1 bash tools/r04_tail.sh c > gpurun_out/r04_fix3_tail.log 2>&1
grep -v "^W2026\|^E2026" gpurun_out/fit_tail_c.log | tail -3
