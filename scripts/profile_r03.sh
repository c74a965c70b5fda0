#!/bin/bash
# On the GPU box, state round 3 ends on: rocprofv3 --kernel-trace --stats of the default bench.py (DDPG headline + the
# sac / naf / kl sub-records), the same command without the profiler, then the counter passes of the three update kernels
# (scripts/profile_r03_pmc.sh).  Outputs under gpurun_out/r03_prof/ and gpurun_out/r03_pmc_*/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03_prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
python scripts/rocpd_extract.py --db $(find $OUT/stats -name "*_results.db" | head -1) --stats --out $OUT/r03_bench_default || exit 1
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1
python -c "
import json
d = json.loads(open('$OUT/bench_plain.json').read().strip().splitlines()[-1])
print('ddpg', d['value'], d['roofline']['frac'], '| sac', d['sac']['value'], d['sac']['roofline']['frac'], '| naf', d['naf']['value'], d['naf']['roofline']['frac'], '| kl', d['kl']['value'], d['kl']['roofline']['frac'], d['kl']['kernel'])
print('cpu', d['cpu_baseline'])
"
for a in ddpg sac naf; do scripts/profile_r03_pmc.sh $a r03 > $OUT/pmc_$a.log 2>&1 || exit 1; tail -3 $OUT/pmc_$a.log; done
