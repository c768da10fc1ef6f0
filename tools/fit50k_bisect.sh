# which earlier leg of bench.py slows the 50 000-sample fit of a 100-step run (7.9 vs 6.5 ms per minibatch)?  bash tools/fit50k_bisect.sh
mkdir -p gpurun_out
out=gpurun_out/fit50k_bisect.log; : > $out
sets=("--steps 20 --warmup 3" "--steps 100" "--steps 100 --no-predictive")
for flags in "${sets[@]}"; do
  timeout 300 python bench.py $flags --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['fit_50k']
print('$flags', '| ms/step', round(d['ms_per_step'],3), '| fit_50k accumulate_s', round(f['accumulate_s'],3), 'host_loop_s', round(f.get('host_loop_s',-1),3), 'W', f.get('power',{}).get('socket_w_median'), 'GHz', f.get('power',{}).get('sclk_ghz_median'), f.get('allocator'))" >> $out
done
cat $out
