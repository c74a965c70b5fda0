#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s24
ABOUT=r03_s24/ab.txt REPS=2 scripts/ab_run2.sh "$@"
