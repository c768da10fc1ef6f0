export TMPDIR=/tmp
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
import ctypes
"
for e in "LK_LANE_PRIO=0" "LK_LANE_PRIO=-1" "LK_LANE_PRIO=-2" "LK_LANE_PRIO=-1 LK_LANES=3" "LK_LANE_PRIO=-1"; do
  echo "[$e] steps: $(env $e timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)  $(env $e timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-predictive --no-eigh --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K20', round(d['ms_per_step'],3))")"
done
