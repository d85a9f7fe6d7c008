#!/bin/bash
# Second round of the search: rings of rows up to MOPT_ICP_LOCK_RINGS in lock step, the rest row by row.
for lr in 1 2 3 8; do
  echo "##### MOPT_ICP_LOCK_RINGS=$lr"
  for d in 2 4 16; do MOPT_ICP_LOCK_RINGS=$lr python3 scripts/icp_offsets_timing.py --per-cell $d --offsets 0.2,0.4,0.7,1.5 2>/dev/null; done
  for k in 4 8 16; do MOPT_ICP_LOCK_RINGS=$lr python3 scripts/icp_offsets_timing.py --surface --radius-spacings $k --offsets 0.1,0.3,0.7 2>/dev/null; done
done
