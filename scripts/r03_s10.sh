#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s10
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
grep -q "rc=0" $OUT/gpu_tests.log || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash scripts/profile_r03.sh
