#!/bin/bash
# A/B of the library's command-batch bound (MOPT_MARKER_EVERY, c_abi.cpp boundCommandBatch): steady-state
# step times and the steps after a synchronisation, same box, same process layout.
for every in 1000000000 8 32 128; do
  echo "##### MOPT_MARKER_EVERY=$every"
  MOPT_MARKER_EVERY=$every PROBE_SET=3 PROBE_PERIOD=1000 python3 scripts/probe_sync_effect.py camera p2p1m p2p10m 2>&1 | grep -v amdgpu.ids
done
