#!/bin/bash
# Whole ICP solves on a scanned surface (1 M x 1 M) under three caps on the grid's cell count.
for lg in 24 26 28; do
  echo "##### MOPT_ICP_MAX_CELLS_LOG2=$lg"
  for k in 2 4 8; do for sh in 0.2 0.6; do
    MOPT_ICP_MAX_CELLS_LOG2=$lg python3 scripts/icp_solve_timing.py --surface --max-dist-spacings $k --shift-radii $sh 2>/dev/null
  done; done
done
