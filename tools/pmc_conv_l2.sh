# PMC pass over the split-fp16 convolution on the c4 sweep shapes: L2 (TCC) requests, hits, misses, bytes to the fabric.
TAG=${1:-x}
export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_l2
cd /tmp && CFGS=2 timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_l2 -o c -- python $GRAFT_REPO_ROOT/tools/conv_f16x2_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_l2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py gpurun_out/pmc_conv_l2_$TAG.md $(find gpurun_out/pmc_l2 -name "*.db") > /dev/null 2>&1
rm -rf gpurun_out/pmc_l2
cat gpurun_out/pmc_conv_l2_$TAG.md | cut -c1-300; tail -3 gpurun_out/pmc_l2.log | cut -c1-300
