# PMC passes over tools/winp_one.py (one fused backward-data shape on the persistent window kernel): L2 hits / misses / fabric
# reads, wave wait buckets.   usage: bash tools/winp_pmc.sh <tag> "<Ci Co H N>" <config>   -> gpurun_out/winp_pmc_<tag>.md
TAG=$1; SHAPE=$2; CFG=${3:-2}
export TMPDIR=/tmp
i=0
DBS=""
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  D=/tmp/wpmc_${TAG}_$i; rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $D -o c -- python $GRAFT_REPO_ROOT/tools/winp_one.py $SHAPE $CFG 6 > /tmp/wpmc_$TAG.log 2>&1)
  DBS="$DBS $(find $D -name '*.db' | head -1)"
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py gpurun_out/winp_pmc_$TAG.md $DBS > /dev/null 2>&1
grep -a "winp\|kernel" gpurun_out/winp_pmc_$TAG.md | cut -c1-400
