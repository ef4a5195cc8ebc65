#!/bin/bash
# TEST INFRASTRUCTURE.  Stages a read-only copy of the reference's Python package under oracle/_ref/reference/ so that a `gpurun` call can run
# the REAL reference on the GPU through the plugin (tests/test_plugin_reference.py).  oracle/_ref/ is git-ignored (the copy
# never enters history) but not gpurun-ignored (it travels to the GPU box with the snapshot).  Usage:
#     oracle/stage_reference.sh            # copy /root/reference/lightx2v -> oracle/_ref/reference/lightx2v
#     X2V_REFERENCE_ROOT=$PWD/oracle/_ref/reference python -m pytest tests/test_plugin_reference.py -m gpu
#     oracle/stage_reference.sh --remove   # drop the copy again (done right after the GPU call: the tree keeps no reference sources)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--remove" ]; then rm -rf oracle/_ref/reference; echo "removed oracle/_ref/reference"; exit 0; fi
SRC=${X2V_REFERENCE_SRC:-/root/reference}
[ -d "$SRC/lightx2v" ] || { echo "no reference at $SRC"; exit 1; }
mkdir -p oracle/_ref/reference
rm -rf oracle/_ref/reference/lightx2v
cp -r "$SRC/lightx2v" oracle/_ref/reference/lightx2v
find oracle/_ref/reference -name "__pycache__" -type d -exec rm -rf {} + 2>/dev/null || true
du -sh oracle/_ref/reference
