#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s14
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/gpu_tests.log
tail -30 $OUT/gpu_tests.log
