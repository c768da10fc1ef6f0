#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/torch_prof_step.py 2>&1 | grep -v "Cannot find" > gpurun_out/torch_prof.log
echo rc=${PIPESTATUS[0]}
grep "COPY" gpurun_out/torch_prof.log | head -30 | cut -c1-400
