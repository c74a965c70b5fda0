#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s29
rm -rf $OUT && mkdir -p $OUT
RLCONTROL_HIP_LIB=$PWD/ab/base.so timeout -k 10 400 python -m pytest tests/test_gpu_ddpg.py -x -q -m gpu -k "mfma and (ten_updates or k_updates or independent or philox_minibatches or learns_critic)" > $OUT/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc" | tee -a $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
[ $rc -eq 0 ] || exit 1
ABOUT=r03_s29/ab.txt REPS=2 scripts/ab_run2.sh prev base
