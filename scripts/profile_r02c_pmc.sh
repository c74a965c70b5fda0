#!/bin/bash
# On the GPU box, final state of round 2: the SQ counter passes of the DDPG headline kernel (separate --pmc runs, program
# directly after `--`, no trace domains) -> gpurun_out/r02c_pmc/*.csv; scripts/sq_summary.py turns them into
# profiles/r02c_mfma_sq_counters.json.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02c_pmc
rm -rf $OUT && mkdir -p $OUT
SHORT="--no-cpu-baseline --no-side-records --steps 4 --warmup 1 --updates-per-step 32"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d $OUT/sq1 -o sq1 -- python3 bench.py $SHORT > $OUT/sq1.json 2> $OUT/sq1.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq2 -o sq2 -- python3 bench.py $SHORT > $OUT/sq2.json 2> $OUT/sq2.err || exit 1
for p in sq1 sq2; do
  python scripts/rocpd_extract.py --db $(find $OUT/$p -name "*_results.db" | head -1) --pmc --out $OUT/r02c_mfma_pmc_$p || exit 1
done
ls $OUT/*.csv
python scripts/sq_summary.py --csv $OUT/r02c_mfma_pmc_sq1_counter_collection.csv --csv $OUT/r02c_mfma_pmc_sq2_counter_collection.csv --updates-per-launch 32 --agents 256 --kernel-us-per-update 296 --tag r02c_mfma && cp profiles/r02c_mfma_sq_counters.json $OUT/
