#!/bin/bash
# What the cap on the number of grid cells (2^24: a 64 MB offset table) costs a scanned surface, whose
# bounding box is mostly empty: MOPT_ICP_MAX_CELLS_LOG2 = 24 / 26 / 28, radius of 2 / 4 / 8 point spacings.
for lg in 24 26 28; do
  echo "##### MOPT_ICP_MAX_CELLS_LOG2=$lg"
  for k in 2 4 8; do
    MOPT_ICP_MAX_CELLS_LOG2=$lg python3 scripts/icp_offsets_timing.py --surface --radius-spacings $k --offsets 0,0.3,0.7 2>/dev/null
  done
done
