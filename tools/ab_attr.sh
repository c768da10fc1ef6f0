# A/B of a class attribute on the same box: bash tools/ab_attr.sh <Class.attr> <value A> <value B> [bench flags]   (A B A B)
#   <Class.attr>: HipKernels.use_pixpair13, KronAccumulator.direct_stack, SplitSweep.fuse_vjp, ...
mkdir -p gpurun_out
out=gpurun_out/ab_attr.log; : > $out
ATTR=$1; A=$2; B=$3; shift 3
FLAGS=${@:---steps 100 --no-cpu-baseline --no-predictive --no-extras --no-check}
for v in $A $B $A $B; do
  timeout 300 python -c "
import sys, runpy
import laplace_amd._lib as L, laplace_amd.backend as Bk, laplace_amd.sweep_nhwc as Sw
ns = {'HipKernels': L.HipKernels, 'KronAccumulator': Bk.KronAccumulator, 'SplitSweep': Sw.SplitSweep, 'HipGGN': Bk.HipGGN}
cls, attr = '$ATTR'.split('.')
setattr(ns[cls], attr, $v)
sys.argv = ['bench.py'] + '$FLAGS'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$ATTR = $v', '| ms/step', round(d['ms_per_step'],3), '| samples/s', round(d['value']), '|', {k:round(v['ms_per_step'],3) for k,v in d['roofline_families'].items() if isinstance(v,dict)})" >> $out
done
cat $out
