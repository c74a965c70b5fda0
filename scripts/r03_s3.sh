#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_s3
ABOUT=r03_s3/ab.txt REPS=2 scripts/ab_run2.sh noexact exact ntx ntt nts30 nts60 nts100
scripts/profile_r03_pmc.sh sac r03a ab/ntx.so > gpurun_out/r03_s3/pmc_sac.log 2>&1; tail -5 gpurun_out/r03_s3/pmc_sac.log
scripts/profile_r03_pmc.sh naf r03a ab/ntx.so > gpurun_out/r03_s3/pmc_naf.log 2>&1; tail -5 gpurun_out/r03_s3/pmc_naf.log
