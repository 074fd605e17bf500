#!/bin/bash
# Build libtamd.so of an EARLIER commit into tools/ab/libtamd_base.so (git-ignored; it travels with the tree to the GPU box),
# for side-by-side A/B through the C ABI (tools/attn_lib_ab.py): no second code arm in the sources.
#   tools/build_base_lib.sh [commit]        default: 23d1524 (the tree round 3 opened its second half with: per-element dropout
#                                           hash, attention without the srcC chains, ABI 6)
# CPU only (hipcc cross-compiles gfx950); ~3 minutes.
set -e
C=${1:-23d1524}
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/tamd_base.XXXXXX)
git -C "$R" archive "$C" transformers_amd include | tar -x -C "$D"
( cd "$D" && python -c "
import sys; sys.path.insert(0, '.')
from transformers_amd import build
print(build._build(build.LIB, build.SOURCES, build.OBJ_DIR, (), True, False))" )
mkdir -p "$R/tools/ab"
cp "$D/transformers_amd/libtamd.so" "$R/tools/ab/libtamd_base.so"
rm -rf "$D"
ls -la "$R/tools/ab/libtamd_base.so"
