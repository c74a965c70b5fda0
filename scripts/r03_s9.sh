#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s9
ABOUT=r03_s9/ab.txt REPS=2 ALGOS="ddpg naf" scripts/ab_run2.sh lbar sep cur2 fuse
