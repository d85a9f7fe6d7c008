import os
for k, v in sorted(os.environ.items()):
    if any(t in k.upper() for t in ("ROCP", "ROCPROF", "HSA_TOOLS", "LD_PRELOAD", "ROCTX")):
        print(k, "=", v[:200])
