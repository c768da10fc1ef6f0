# A/B of two builds of the library on the same box: bash tools/ab_libs.sh <variant .so> [bench flags]   (the default build runs first and last)
mkdir -p gpurun_out
out=gpurun_out/ab_libs.log; : > $out
V=$1; shift
FLAGS=${@:---steps 100 --no-cpu-baseline --no-predictive --no-extras --no-check}
for lib in "" "$V" "" "$V"; do
  LK_LIB=$lib timeout 300 python bench.py $FLAGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('${lib:-default}', '| ms/step', round(d['ms_per_step'],3), '| samples/s', round(d['value']), '| winp us', round(r['avg_launch_ms']*1e3,1), 'frac', round(r['frac'],3), '|', {k:round(v['ms_per_step'],3) for k,v in d['roofline_families'].items() if isinstance(v,dict)})" >> $out
done
cat $out
