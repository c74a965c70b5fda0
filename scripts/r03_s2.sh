#!/bin/bash
# GPU session 2 of round 3: whole GPU suite on the new library, any-shape kernel throughput at 512 threads, variant A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s2
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
grep -q "rc=0" $OUT/gpu_tests.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-side-records --kernel generic --updates-per-step 16 --steps 5 --warmup 1 > $OUT/bench_generic.json 2> $OUT/bench_generic.err
python -c "import json; d=json.loads(open('$OUT/bench_generic.json').read().strip().splitlines()[-1]); print('ddpg generic', round(d['value']))"
timeout -k 10 300 python scripts/bench_sac_naf.py --kernel generic --tag r03_generic --records 100000 > $OUT/sac_naf_generic.log 2>&1; tail -3 $OUT/sac_naf_generic.log
timeout -k 10 300 python scripts/dropin_latency.py --only DDPG --tag r03 > $OUT/dropin.log 2>&1; tail -6 $OUT/dropin.log
ABOUT=r03_s2/ab.txt REPS=1 scripts/ab_run2.sh base nt prio a11 a12 a13 stag stag8 ntstag base
for na in 1 32 128; do
  v=$(RLCONTROL_HIP_LIB=$PWD/ab/base.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-side-records --agents $na --updates-per-step 64 --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])")
  echo "agents $na ddpg=$v" | tee -a $OUT/ab.txt
done
