#!/bin/bash
# Same-box A/B of two builds of libphendiff_hip.so on the per-layer forward profile (box-to-box spread is +-3 %, larger than
# most kernel changes).  Alternative builds live under build_ab/ at the repo root (git-ignored, shipped by gpurun), never inside
# the package:   PD_LIB_OLD=build_ab/<old>.so scripts/ab_forward.sh
for i in 1 2; do
  PD_LIB=$PD_LIB_OLD python scripts/profile_forward.py 2>/dev/null | grep -E "^total" | sed 's/^/old: /'
  python scripts/profile_forward.py 2>/dev/null > /tmp/new_prof.txt; grep -E "^total" /tmp/new_prof.txt | sed 's/^/new: /'
done
