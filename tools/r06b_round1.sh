#!/bin/bash
# round 6 (second session), first look at the LDS-broadcast factorisation of k_reduced: parity, cfg3 bench + timeline, per-rank sizes, phase stamps
set -u
R=$PWD; O=$R/gpurun_out/${1:-r06b_1}; mkdir -p $O
bash tools/quick_round.sh $(basename $O)
for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do
  echo "== reduced stamps $spec" >> $O/reduced_stamps.txt
  VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec >> $O/reduced_stamps.txt 2>&1
done
cat $O/reduced_stamps.txt
