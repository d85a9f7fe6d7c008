#!/usr/bin/env python3
"""Reads the gfx950 code-object metadata of the library's own kernels and fails on any register
spill or scratch use (a short list of one-workgroup step kernels with accepted scalar-register spills
aside: ACCEPTED_SGPR_SPILLS below, each with its bound and its reason).

    python scripts/check_spills.py            # every .hip source of the library, Makefile flags
    python scripts/check_spills.py --table    # also print one line per kernel
    python scripts/check_spills.py --only reproj

Each source is compiled device-only with exactly the flags the Makefile gives it (asked of `make -n`),
the gfx950 code object is unbundled and `llvm-readelf --notes` is parsed for `.sgpr_spill_count`,
`.vgpr_spill_count` and `.private_segment_fixed_size` per kernel.  hipcc cross-compiles, so this runs
without a GPU; tests/test_code_objects.py runs it in the CPU suite.
"""
import argparse
import os
import re
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
DEVICE_SOURCES = ("sweep_kernels", "fd_kernels", "icp_grid", "lm_kernels")
# Known and accepted: (kernel-name prefix, most scalar registers it may spill, why).  Nothing here may
# spill a VGPR or use scratch, and every kernel that streams data per element must stay off this list.
ACCEPTED_SGPR_SPILLS = (
    # one workgroup, once per LM step: what is left after round 5 took the pivoted LDL^T's predicated
    # exchanges out of the step (the pivoted solve runs from LDS, cold: lm_device.hpp proposeTrial) —
    # 118 -> 18 in lmStepKernel<double>, 52 -> 0 in the fused finalize-and-step kernels
    ("mopt::lmStepKernel<", 24, "LM step body"),
    # fp32, symmetric covariance, arguments from HBM: <= 20 values (partials pointer, counts, the
    # set's block bounds) parked in a VGPR's lanes once per launch, outside the tile loop
    ("mopt::p2pForwardDiffResidentArgsKernel<float,", 20, "parked outside the tile loop"),
    ("mopt::p2pForwardDiffResidentSetKernel<float,", 20, "parked outside the tile loop"),
    # the same body behind the per-point choice of the device-resident loop (round 6): 17
    ("mopt::p2pForwardDiffEitherResidentKernel<float,", 20, "parked outside the tile loop"),
    # one workgroup of 256 threads running a whole small minimisation (512 registers a lane): the LM
    # step inside a loop keeps more lane masks and invariants alive than the one-shot kernels do
    # (the same step there: 0-18) — parked in vector lanes.  330 with the parameter count read at run
    # time, 92 with n = 6 fixed at compile time (the solve's dispatch on n and every `i < n` predicate
    # fold away); a point costs 7-8 us instead of 10-11 (profiles/r5_small_solve.txt)
    ("mopt::p2pSolveSmallKernel<", 100, "LM step inside a loop"),
    # the variants that also hold the literal forward-difference form for a cost whose sweep is chosen per
    # point (round 6; FD_COV = 0, 1, 2): the fp64 ones park 160 / 160 / 208 scalars — the same step plus
    # the choice and a second sweep body in the loop; measured +0.3-0.4 us per evaluated point against the
    # moments-only variant (profiles/r6_small_solve.txt), against +3-4 us for the launch-per-point loop
    ("mopt::p2pSolveSmallKernel<double, 0>", 170, "LM step + two sweep forms inside a loop"),
    ("mopt::p2pSolveSmallKernel<double, 1>", 170, "LM step + two sweep forms inside a loop"),
    ("mopt::p2pSolveSmallKernel<double, 2>", 216, "LM step + two sweep forms inside a loop"),
)
# Vector registers the allocator moved to the accumulation registers of the same lane (vgpr spills
# with NO scratch memory): for the kernel that holds its correspondences in registers across a whole
# minimisation that is where idle values belong.
ACCEPTED_AGPR_RESIDENT = (
    ("mopt::p2pSolveSmallKernel<", 32),  # (0 since n is fixed at compile time; 20-28 before)
)
FIELDS = ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")


def compile_command(stem):
    """The Makefile's own command line for build/obj/<stem>.o (so a flag added there is checked here)."""
    out = subprocess.check_output(["make", "-C", ROOT, "-n", "-B", "build/obj/%s.o" % stem], text=True)
    for line in out.splitlines():
        if "hipcc" in line and ("%s.hip" % stem) in line:
            return shlex.split(line)
    raise RuntimeError("no hipcc line for %s in `make -n`" % stem)


def kernel_metadata(stem, workdir):
    cmd = compile_command(stem)
    bundle = os.path.join(workdir, stem + ".bundle")
    elf = os.path.join(workdir, stem + ".elf")
    i = cmd.index("-o")
    cmd[i + 1] = bundle
    cmd.insert(1, "--offload-device-only")
    subprocess.check_call(cmd, cwd=ROOT)
    subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--input=" + bundle, "--targets=" + TARGET, "--output=" + elf])
    notes = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", elf], text=True)
    kernels, current = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, value = m.groups()
        if key == "name" and value.startswith("_Z"):
            # '.name' also labels kernel arguments; kernel records carry a mangled name and .sgpr_count
            current = {"name": value}
            kernels.append(current)
        elif current is not None and key in FIELDS:
            current[key] = int(value)
    kernels = [k for k in kernels if "sgpr_count" in k]
    # kernel arguments as the code object lays them out: (offset, size, value_kind) per kernel, for the
    # check of the direct-dispatch path's own packing (csrc/aql.cpp)
    by_name = {k["name"]: k for k in kernels}
    for record in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        names = [n for n in re.findall(r"\.name:\s+(_Z\S+)", record) if n in by_name]
        if not names:
            continue
        args = re.findall(r"\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)\n\s+\.value_kind:\s+(\S+)", record)
        by_name[names[-1]]["args"] = [(int(o), int(z), kind) for o, z, kind in args]
        m = re.search(r"\.kernarg_segment_size:\s+(\d+)", record)
        by_name[names[-1]]["kernarg_segment_size"] = int(m.group(1)) if m else 0
    return kernels


# code object v5 implicit arguments, relative to the first one, as csrc/aql.cpp fills them
IMPLICIT_LAYOUT = {"hidden_block_count_x": 0, "hidden_block_count_y": 4, "hidden_block_count_z": 8,
                   "hidden_group_size_x": 12, "hidden_group_size_y": 14, "hidden_group_size_z": 16,
                   "hidden_remainder_x": 18, "hidden_remainder_y": 20, "hidden_remainder_z": 22,
                   "hidden_global_offset_x": 40, "hidden_global_offset_y": 48, "hidden_global_offset_z": 56,
                   "hidden_grid_dims": 64}
# implicit arguments a dispatcher would have to provide real objects for: no kernel of the library may use them
IMPLICIT_UNSUPPORTED = ("hidden_printf_buffer", "hidden_hostcall_buffer", "hidden_multigrid_sync_arg",
                        "hidden_heap_v1", "hidden_default_queue", "hidden_completion_action",
                        "hidden_dynamic_lds_size", "hidden_queue_ptr")


def check_kernarg_layout(kernel):
    """Problems (strings) with a kernel's argument layout as the direct-dispatch path assumes it: explicit
    arguments first, each at its natural alignment (nothing to check: that is the ABI), then — from the
    next 8-byte boundary — the implicit block in IMPLICIT_LAYOUT's order; everything inside one
    4 KB argument slot."""
    problems = []
    args = kernel.get("args", [])
    explicit_end, first_hidden = 0, None
    for offset, size, kind in args:
        if kind.startswith("hidden_"):
            if first_hidden is None:
                first_hidden = offset
        else:
            if first_hidden is not None:
                problems.append("explicit argument behind an implicit one")
            explicit_end = max(explicit_end, offset + size)
    if first_hidden is not None:
        if first_hidden != (explicit_end + 7) // 8 * 8:
            problems.append("implicit block at %d, explicit arguments end at %d" % (first_hidden, explicit_end))
        for offset, size, kind in args:
            if kind in IMPLICIT_LAYOUT and offset - first_hidden != IMPLICIT_LAYOUT[kind]:
                problems.append("%s at +%d, expected +%d" % (kind, offset - first_hidden, IMPLICIT_LAYOUT[kind]))
            if kind in IMPLICIT_UNSUPPORTED:
                problems.append("uses %s" % kind)
    if kernel.get("kernarg_segment_size", 0) > 4096 or explicit_end > 3072:
        problems.append("argument segment of %d bytes (explicit %d)" % (kernel.get("kernarg_segment_size", 0), explicit_end))
    return problems


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True, check=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "")) for n in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", action="store_true", help="print every kernel, not only offenders")
    ap.add_argument("--only", default="", help="substring filter on the demangled kernel name")
    ap.add_argument("--sources", nargs="*", default=list(DEVICE_SOURCES))
    args = ap.parse_args()
    bad = total = accepted = 0
    with tempfile.TemporaryDirectory() as workdir:
        for stem in args.sources:
            kernels = kernel_metadata(stem, workdir)
            names = demangle([k["name"] for k in kernels])
            for k, name in zip(kernels, names):
                if args.only and args.only not in name:
                    continue
                total += 1
                sgpr_spills = k.get("sgpr_spill_count", 0)
                in_agprs = max([limit for prefix, limit in ACCEPTED_AGPR_RESIDENT if name.startswith(prefix)],
                               default=0)
                scratch = k.get("private_segment_fixed_size", 0)
                hard = scratch or k.get("vgpr_spill_count", 0) > in_agprs
                allowed = max([limit for prefix, limit, _ in ACCEPTED_SGPR_SPILLS if name.startswith(prefix)],
                              default=0)
                status = "ok"
                if hard or sgpr_spills > allowed:
                    status = "SPILL"
                    bad += 1
                elif sgpr_spills or k.get("vgpr_spill_count", 0):
                    status = "known"
                    accepted += 1
                for problem in check_kernarg_layout(k):
                    # (every kernel is checked; the direct path dispatches the blocking sweeps and finalizes)
                    print("KERNARG %s.hip  %s: %s" % (stem, name, problem))
                    bad += 1
                if status != "ok" or args.table:
                    print("%-6s %s.hip  sgpr %3d (spill %3d)  vgpr %3d (spill %3d)  scratch %4d  %s" % (
                        status, stem, k["sgpr_count"], sgpr_spills, k["vgpr_count"],
                        k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0), name))
    print("%d kernels checked: %d with spills or scratch, %d with accepted scalar-register spills "
          "(ACCEPTED_SGPR_SPILLS)" % (total, bad, accepted))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
