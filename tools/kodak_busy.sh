#!/bin/bash
# Development aid: is the Kodak leg GPU-bound?  Kernel trace of tools/kodak_fit.py (ITERS iterations, modes in MODES),
# reduced on the box to busy / idle figures (tools/busy_union.py).  COUNT images (default 24).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for g in ${MODES:-1 3}; do
  rm -rf /tmp/kb_$g
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kb_$g -o run -- python3 $R/tools/kodak_fit.py ${COUNT:-24} ${ITERS:-10000} $g 2>&1 | grep "mode"
  python3 $R/tools/busy_union.py /tmp/kb_$g
done
