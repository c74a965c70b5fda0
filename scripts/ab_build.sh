#!/bin/bash
# developer loop: build the BASELINE-shape library (RLC_FAST_BUILD=1) and park it as ab/<name>.so for A/B runs on one box
#   scripts/ab_build.sh <name> [extra hipcc flags via RLC_EXTRA_CFLAGS]
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
RLC_FAST_BUILD=1 python rlcontrol_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -30 /tmp/ab_build.log; exit 1; }
cp rlcontrol_amd/librlcontrol_hip.so ab/$1.so
echo "ab/$1.so"
