#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s19
mkdir -p $OUT
for v in base nont ws; do
  echo "== $v" | tee -a $OUT/dropin.log
  RLCONTROL_HIP_LIB=$PWD/ab/$v.so timeout -k 10 300 python scripts/dropin_latency.py --only DDPG,SoftActorCritic --batches 100 --no-split --tag r03s19_$v 2>&1 | tee -a $OUT/dropin.log
done
