#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s4
ABOUT=r03_s4/ab.txt REPS=2 scripts/ab_run2.sh nont cur noearly st20 st40 st60 st100
