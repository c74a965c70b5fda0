#!/bin/bash
# GPU session 1 of round 3: RCCL rehearsal tests, measured peaks + counter calibration, timing ablations
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_s1
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu > $OUT/rccl_test.log 2>&1; echo "rccl rc=$?" | tee -a $OUT/rccl_test.log
tail -3 $OUT/rccl_test.log
scripts/micro/peaks > $OUT/peaks.json 2> $OUT/peaks.err && cat $OUT/peaks.json
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf -- scripts/micro/peaks 536870912 > $OUT/peaks_f.json 2> $OUT/pf.err && \
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw -- scripts/micro/peaks 536870912 > $OUT/peaks_w.json 2> $OUT/pw.err && \
for p in pf pw; do python scripts/rocpd_extract.py --db $(find $OUT/$p -name "*_results.db" | head -1) --pmc --kernel "" --out $OUT/peaks_$p; done
python scripts/peaks_summary.py --plain $OUT/peaks_f.json --fetch $OUT/peaks_pf_counter_collection.csv --write $OUT/peaks_pw_counter_collection.csv --tag r03_calib > $OUT/calib.log 2>&1; tail -30 $OUT/calib.log
ABOUT=r03_s1/ablate.txt REPS=1 scripts/ab_run2.sh base a0 a1 a2 a3 a4 a5 a6 a7 a8 a9 a10 a56 base
