#!/bin/bash
# same-box A/B of two library builds on the cfg3 pass timeline: tools/r06b_ab_tl.sh libA.so libB.so  (kernel durations of the steady passes)
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  VICALIB_AMD_LIB=$PWD/$lib bash tools/timeline_round.sh cfg3 k_final 2>&1 | grep -A17 "^kernel" | tail -17 | awk '{printf "%s %s | ", substr($1,1,22), $3} END {print ""}'
done
done
