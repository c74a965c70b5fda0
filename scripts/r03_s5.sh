#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s5
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
ABOUT=r03_s5/ab.txt REPS=2 scripts/ab_run2.sh nolbar lbar
timeout -k 10 300 python scripts/dropin_latency.py --only DDPG --tag r03b > $OUT/dropin.log 2>&1; tail -6 $OUT/dropin.log
