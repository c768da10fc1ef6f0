#!/bin/bash
# The evidence pass of a round on the MI355X box, on the FINAL kernel sources (the PMC table's stamp must match them):
#   1. the whole -m gpu suite; the parity numbers its rel() helpers measured -> gpurun_out/<R>_parity_errors_<tag>.log
#   2. PMC traffic passes over the bench command (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only)
#   3. rocprofv3 --kernel-trace --stats of the driver's bench command, and of 20 serial steady-state steps
#   4. the bench lines: the driver's `--steps 20 --warmup 5` and the default flags
#   5. smoke()
# Writes under gpurun_out/; copy what is to be judged into profiles/.
# usage: bash tools/gpu_evidence.sh <round, e.g. r05> [tag] [notests] [nodefault]
R5=${1:-r05}; TAG=${2:-v1}
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
S=gpurun_out/${R5}_summary_$TAG.log; : > $S
if [[ " $* " != *" notests "* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -rs --durations=10 > gpurun_out/${R5}_gpu_tests_$TAG.log 2>&1; echo "tests rc=$?" >> $S
  tail -4 gpurun_out/${R5}_gpu_tests_$TAG.log >> $S
  cp gpurun_out/parity_errors.log gpurun_out/${R5}_parity_errors_$TAG.log 2>/dev/null
fi
LIGHT="--no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check"
rm -rf gpurun_out/pmc gpurun_out/prof_$TAG gpurun_out/ks_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 $LIGHT > $R/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $R/$S
  cd $R
done
python tools/pmc_traffic.py $(find gpurun_out/pmc -name "*FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmc -name "*WRITE_SIZE*.db" | head -1) gpurun_out/${R5}_pmc_traffic_bench_c4_$TAG.json gpurun_out/${R5}_pmc_traffic_bench_c4_$TAG.md > /dev/null 2>> $S
rm -rf gpurun_out/pmc
cp gpurun_out/${R5}_pmc_traffic_bench_c4_$TAG.json profiles/   # (so that the bench runs below find a table with a matching stamp)
# PMC traffic of the GLM predictive (the metric's second half): four passes, with 4 calls and with none; the difference is theirs
rm -rf gpurun_out/pmcp
for n in 0 4; do for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcp -o p${n}_$c -- python $R/tools/kron_predictive_c4.py --calls $n > $R/gpurun_out/pmcp_${n}_$c.log 2>&1
  echo "pmc predictive $n $c rc=$?" >> $R/$S
  cd $R
done; done
python tools/pmc_traffic.py --diff $(find gpurun_out/pmcp -name "*p0_FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmcp -name "*p0_WRITE_SIZE*.db" | head -1) $(find gpurun_out/pmcp -name "*p4_FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmcp -name "*p4_WRITE_SIZE*.db" | head -1) 4 gpurun_out/${R5}_pmc_traffic_predictive_c4_$TAG.json gpurun_out/${R5}_pmc_traffic_predictive_c4_$TAG.md > /dev/null 2>> $S
rm -rf gpurun_out/pmcp
cp gpurun_out/${R5}_pmc_traffic_predictive_c4_$TAG.json profiles/
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py --steps 20 --warmup 5 $LIGHT > $R/gpurun_out/prof_$TAG.log 2>&1; echo "trace rc=$?" >> $R/$S
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/${R5}_bench_c4_kernel_stats_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/ks_$TAG -o p -- python $R/tools/steps_only.py 20 serial > $R/gpurun_out/ks_$TAG.log 2>&1; echo "serial steps trace rc=$?" >> $R/$S
cd $R
python tools/rocpd_stats.py $(find gpurun_out/ks_$TAG -name "*.db" | head -1) gpurun_out/${R5}_steps_serial_kernel_stats_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/ks_$TAG
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${R5}_bench_k20_$TAG.log 2>&1; echo "bench k20 rc=$?" >> $S
grep '^{' gpurun_out/${R5}_bench_k20_$TAG.log | tail -1 > gpurun_out/${R5}_bench_c4_k20_$TAG.json
if [[ " $* " != *" nodefault "* ]]; then
  timeout 900 python bench.py > gpurun_out/${R5}_bench_$TAG.log 2>&1; echo "bench rc=$?" >> $S
  grep '^{' gpurun_out/${R5}_bench_$TAG.log | tail -1 > gpurun_out/${R5}_bench_c4_$TAG.json
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $S 2>&1
cat $S; python - <<PY
import json
for f in ("gpurun_out/${R5}_bench_c4_k20_$TAG.json", "gpurun_out/${R5}_bench_c4_$TAG.json"):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        print(f, "missing", e); continue
    r = d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"], 3), "frac", round(r["frac"], 3), "bound", r.get("bound"), "frac_hbm", r.get("frac_hbm"),
          "traffic", r["traffic"], str(r.get("traffic_source"))[:90])
    print("  predictive", d.get("predictive_samples_per_s"), "eigh_ms", d.get("eigh_ms"), "others", {k: round(v.get("fit_ms", 0), 1) for k, v in (d.get("other_configs") or {}).items() if isinstance(v, dict)})
PY
