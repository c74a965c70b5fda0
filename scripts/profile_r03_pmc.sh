#!/bin/bash
# On the GPU box: SQ counter passes + FETCH / WRITE passes of one update kernel (separate --pmc runs, program directly
# after `--`, no trace domains).   scripts/profile_r03_pmc.sh <algo: ddpg|sac|naf> <tag> [lib.so]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ALGO=$1; TAG=$2
[ -n "$3" ] && export RLCONTROL_HIP_LIB=$PWD/$3
OUT=$PWD/gpurun_out/${TAG}_pmc_$ALGO
rm -rf $OUT && mkdir -p $OUT
if [ $ALGO = ddpg ]; then SEL="--no-side-records"; KERN=rlc_ddpg_update_mfma_kernel; else SEL="--side-only $ALGO"; KERN=rlc_${ALGO}_update_mfma_kernel; fi
SHORT="--no-cpu-baseline $SEL --steps 4 --warmup 1 --updates-per-step 32"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES -d $OUT/sq1 -o sq1 -- python3 bench.py $SHORT > $OUT/sq1.json 2> $OUT/sq1.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq2 -o sq2 -- python3 bench.py $SHORT > $OUT/sq2.json 2> $OUT/sq2.err || exit 1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf -- python3 bench.py $SHORT > $OUT/pf.json 2> $OUT/pf.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw -- python3 bench.py $SHORT > $OUT/pw.json 2> $OUT/pw.err || exit 1
for p in sq1 sq2 pf pw; do
  python scripts/rocpd_extract.py --db $(find $OUT/$p -name "*_results.db" | head -1) --pmc --out $OUT/${TAG}_${ALGO}_pmc_$p || exit 1
done
US=$(python -c "import json; d=json.loads(open('$OUT/pf.json').read().strip().splitlines()[-1]); print(256.0/d['value']*1e6)")
python scripts/sq_summary.py --csv $OUT/${TAG}_${ALGO}_pmc_sq1_counter_collection.csv --csv $OUT/${TAG}_${ALGO}_pmc_sq2_counter_collection.csv --updates-per-launch 32 --agents 256 --kernel $KERN --kernel-us-per-update $US --tag ${TAG}_${ALGO} > $OUT/sq_summary.log && cp profiles/${TAG}_${ALGO}_sq_counters.json $OUT/
python scripts/pmc_summary.py --fetch $OUT/${TAG}_${ALGO}_pmc_pf_counter_collection.csv --write $OUT/${TAG}_${ALGO}_pmc_pw_counter_collection.csv --updates-per-launch 32 --agents 256 --kernel $KERN --tag ${TAG}_${ALGO} > $OUT/pmc_summary.log && cp profiles/${TAG}_${ALGO}_pmc_traffic.json $OUT/
tail -12 $OUT/sq_summary.log | head -30
