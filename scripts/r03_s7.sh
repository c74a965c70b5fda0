#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s7
ABOUT=r03_s7/ab.txt REPS=2 scripts/ab_run2.sh lbar s25 s50 s75 s110
