mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "ll_ggn_full_and_quadform" > gpurun_out/t_ll.log 2>&1
echo "tests rc=$?" > gpurun_out/summary_ll.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eigh > gpurun_out/bench_ll.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary_ll.log
tail -3 gpurun_out/t_ll.log; tail -1 gpurun_out/bench_ll.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('predictive'))"; cat gpurun_out/summary_ll.log
