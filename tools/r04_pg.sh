export TMPDIR=/tmp
for pg in 8 5 4; do
  echo "LK_PIX_GROUP=$pg steady: $(LK_PIX_GROUP=$pg timeout 300 python tools/steps_only.py 80 2>&1 | tail -1)"
  LK_PIX_GROUP=$pg timeout 300 python tools/fit_ab.py 20 4 2>&1 | tail -1
  LK_PIX_GROUP=$pg timeout 300 python tools/fit_ab.py 391 1 2>&1 | tail -1
done
