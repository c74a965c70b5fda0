#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s17
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_kl.py tests/test_gpu_rollout.py -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
timeout -k 10 300 python scripts/dropin_latency.py --only ReverseKL,ForwardKL --tag r03d > $OUT/dropin.log 2>&1; tail -8 $OUT/dropin.log
