# in-step A/B on one box (separate processes, alternating): $1 / $2 = environment assignments of the two arms
export TMPDIR=/tmp
for rep in 1 2 3; do
  echo "A [$1]: $(env $1 timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)"
  echo "B [$2]: $(env $2 timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)"
done
