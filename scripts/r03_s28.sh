#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s28
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc" | tee -a $OUT/gpu_tests.log
tail -5 $OUT/gpu_tests.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
bash scripts/profile_r03.sh
