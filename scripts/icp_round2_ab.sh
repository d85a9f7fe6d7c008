#!/bin/bash
# Round 5: the correspondence search's second round in lock step (reach 1).  Offsets of 0 ... 1.5 radii,
# 1 and 4 targets to the radius cube with one cell to the radius (MOPT_ICP_REACH=1), then the default grid.
for d in 1 4; do MOPT_ICP_REACH=1 python3 scripts/icp_offsets_timing.py --per-cell $d --offsets 0,0.2,0.4,0.5,0.7,1.0,1.5 2>/dev/null; done
for d in 1 2 4 16; do python3 scripts/icp_offsets_timing.py --per-cell $d --offsets 0,0.2,0.4,0.7,1.5 2>/dev/null; done
MOPT_ICP_REACH=1 python3 scripts/icp_offsets_timing.py --dtype f32 --per-cell 1 --offsets 0,0.2,0.7,1.5 2>/dev/null
