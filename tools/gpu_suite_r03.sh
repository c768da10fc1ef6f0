# Round-3 GPU pass: the whole -m gpu suite (incl. the reference's own classes on the real kernels, unpacked from
# oracle/_ref/), then the default bench line.  usage: bash tools/gpu_suite_r03.sh [tag] [pytest args]
TAG=${1:-a}
shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=15 "$@" > gpurun_out/r03_tests_$TAG.log 2>&1
echo "tests rc=$?" > gpurun_out/r03_summary_$TAG.log
tail -40 gpurun_out/r03_tests_$TAG.log
timeout 900 python bench.py > gpurun_out/r03_bench_$TAG.log 2>&1
echo "bench rc=$?" >> gpurun_out/r03_summary_$TAG.log
cat gpurun_out/r03_summary_$TAG.log
tail -c 6000 gpurun_out/r03_bench_$TAG.log
