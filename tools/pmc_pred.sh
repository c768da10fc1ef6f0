# PMC passes over one predictive call set (tools/kron_predictive_c4.py --calls 1): per-launch counters of the quadratic-form kernel
export TMPDIR=/tmp
DBS=""
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  D=/tmp/pp_$i; rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $D -o c -- python $GRAFT_REPO_ROOT/tools/kron_predictive_c4.py --calls 0 > /tmp/pp.log 2>&1)
  DBS="$DBS $(find $D -name '*.db' | head -1)"
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python tools/pmc_launches.py quadform_conv_planes $DBS
