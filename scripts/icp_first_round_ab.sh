# the correspondence search with and without its first round, against the distance between the
# clouds, the density of the grid and the scalar type
set -e
for on in 0 1; do
  MOPT_ICP_FIRST_ROUND=$on python scripts/icp_offsets_timing.py 2>/dev/null
  MOPT_ICP_FIRST_ROUND=$on python scripts/icp_offsets_timing.py --per-cell 4 2>/dev/null
  MOPT_ICP_FIRST_ROUND=$on python scripts/icp_offsets_timing.py --dtype f32 2>/dev/null
done
