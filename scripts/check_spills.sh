#!/bin/bash
# Builds every device source of the library with the Makefile's flags and fails on any register spill or
# scratch use in the code objects (scripts/check_spills.py; no GPU needed: hipcc cross-compiles gfx950).
cd "$(dirname "$0")/.." && exec python3 scripts/check_spills.py "$@"
