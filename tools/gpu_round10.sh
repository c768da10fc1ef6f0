#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-20)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$tag -- python $GRAFT_REPO_ROOT/tools/pmc_gram.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
done
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/pmc >> gpurun_out/summary.log
for f in gpurun_out/pmc/*.db; do python tools/rocpd_pmc.py $f > ${f%.db}.txt 2>&1; done
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log | cut -c1-400; cat gpurun_out/summary.log
