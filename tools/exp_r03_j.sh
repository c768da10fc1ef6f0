mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > gpurun_out/sq_counters.txt
wc -w gpurun_out/sq_counters.txt; head -c 3000 gpurun_out/sq_counters.txt
