mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "quadform or logdet" > gpurun_out/t_new.log 2>&1
echo "new kernel tests rc=$?" > gpurun_out/summary_new.log
timeout 900 python -m pytest tests/test_gpu_backend.py tests/test_dict_inputs_c5.py tests/test_laplace_e2e.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_backend.log 2>&1
echo "backend tests rc=$?" >> gpurun_out/summary_new.log
timeout 600 python tools/marglik_bench.py > gpurun_out/marglik.log 2>&1
echo "marglik rc=$?" >> gpurun_out/summary_new.log
timeout 900 python tools/c5_bert.py > gpurun_out/c5.log 2>&1
echo "c5 rc=$?" >> gpurun_out/summary_new.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_new.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary_new.log
tail -3 gpurun_out/t_new.log; tail -3 gpurun_out/t_backend.log; tail -1 gpurun_out/marglik.log; tail -1 gpurun_out/c5.log | cut -c1-900; tail -1 gpurun_out/bench_new.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','eigh_ms','predictive','predictive_kron_c4')})"; cat gpurun_out/summary_new.log
