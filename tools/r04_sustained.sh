export TMPDIR=/tmp; mkdir -p gpurun_out
( for i in $(seq 1 14); do sleep 8; echo "smi t=$((i*8))s $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i 'sclk clock\|Package Power\|junction' | sed 's/GPU\[0\]\s*:\s*//' | tr '\n' ';')"; done ) > gpurun_out/r04_sustained_smi.log 2>&1 &
python tools/sustained.py 34 400 2>&1 | grep chunk > gpurun_out/r04_sustained.log
wait
awk 'NR%3==1' gpurun_out/r04_sustained.log; cat gpurun_out/r04_sustained_smi.log
