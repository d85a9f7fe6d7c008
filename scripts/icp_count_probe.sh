set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/icp_count
cat > gpurun_out/icp_count/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import moptimizer_0_amd as mo
n = 1_000_000
rng = np.random.default_rng(1)
tgt = rng.random((n, 3)) * 100.0
src = tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3)) + np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0) * 0.2
dt = np.float32 if os.environ.get("ICP_F32") else np.float64
cost = mo.IcpCost(src, tgt, 1.0, dtype=dt)
count = os.environ.get("ICP_COUNT", "1") == "1"
for _ in range(10):
    cost.update(np.zeros(6), count_matches=count)
cost.synchronize()
cost.close()
PY
for c in 1 0; do
  ICP_COUNT=$c timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/icp_count/c$c -o s -- python3 gpurun_out/icp_count/run.py > /dev/null 2> gpurun_out/icp_count/c$c.err
  echo "count=$c"; grep icpMatch gpurun_out/icp_count/c$c/*/s_kernel_stats.csv gpurun_out/icp_count/c$c/s_kernel_stats.csv 2>/dev/null | cut -c1-200
done
