#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s23
ABOUT=r03_s23/ab.txt REPS=2 ALGOS="ddpg naf" scripts/ab_run2.sh prev base ex
