#!/bin/bash
# round 6 (second session): hybrid factorisation of k_reduced + matrix-pipe J^T J of k_imu_jac: parity, bench, timeline, per-rank sizes, stamps
set -u
R=$PWD; O=$R/gpurun_out/${1:-r06b_2}; mkdir -p $O
bash tools/quick_round.sh $(basename $O)
for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do
  echo "== reduced stamps $spec" >> $O/reduced_stamps.txt
  VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec >> $O/reduced_stamps.txt 2>&1
done
cat $O/reduced_stamps.txt
VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_ijstamps.so python tools/ij_stamps.py cfg3 > $O/ij_stamps.txt 2>&1; cat $O/ij_stamps.txt
