#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gram_trace.py 2>&1 | grep -v "Cannot find" > gpurun_out/gram_trace.log
echo "rc=${PIPESTATUS[0]}"
cat gpurun_out/gram_trace.log
