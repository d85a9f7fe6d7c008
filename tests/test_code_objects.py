"""The library's own gfx950 code objects carry no register spills and no scratch (hipcc cross-compiles:
no GPU needed).  scripts/check_spills.py compiles every device source with the Makefile's flags and reads
the metadata of every kernel; the one-workgroup LM step kernels are on its short accepted list."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_spills_registers_or_uses_scratch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spills.py")],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert " 0 with spills or scratch" in last, last
    # every reprojection sweep (BASELINE config 5) is among the checked kernels and clean
    table = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spills.py"), "--table",
                            "--sources", "sweep_kernels", "--only", "reproj"],
                           capture_output=True, text=True, timeout=900)
    rows = [ln for ln in table.stdout.splitlines() if "mopt::reproj" in ln]
    assert len(rows) >= 10 and all(ln.startswith("ok") for ln in rows), table.stdout[-3000:]
