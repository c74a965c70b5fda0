#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s6
ABOUT=r03_s6/ab.txt REPS=2 scripts/ab_run2.sh lbar b2 b2nostage b2nopack
