#!/bin/bash
# on the GPU box: bench every ab/<name>.so given on the command line, twice, interleaved (same box, same session)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for n in "$@"; do
    v=$(RLCONTROL_HIP_LIB=$PWD/ab/$n.so python bench.py --no-cpu-baseline --no-side-records --updates-per-step 64 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
    echo "$n rep$rep $v"
  done
done
