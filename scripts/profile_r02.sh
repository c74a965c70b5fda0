#!/bin/bash
# On the GPU box: rocprofv3 passes of bench.py for the round's profiles (kernel stats of the default run, then separate
# --pmc passes on a short run).  Program directly after `--`; counters never combined with trace domains other than
# --kernel-trace.  Outputs under gpurun_out/r02_prof/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02_prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
SHORT="--no-cpu-baseline --no-side-records --steps 4 --warmup 1 --updates-per-step 32"
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 bench.py $SHORT > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python3 bench.py $SHORT > $OUT/write.json 2> $OUT/write.err || exit 1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d $OUT/sq1 -o sq1 -- python3 bench.py $SHORT > $OUT/sq1.json 2> $OUT/sq1.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq2 -o sq2 -- python3 bench.py $SHORT > $OUT/sq2.json 2> $OUT/sq2.err || exit 1
python bench.py --no-cpu-baseline --no-side-records > $OUT/bench_plain.json 2>/dev/null
find $OUT -name "*.csv" | head -40
