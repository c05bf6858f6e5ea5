#!/bin/bash
# local helper: rebuild the extension, and only if that worked send the given command to the GPU box
cd /root/repo || exit 1
python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { grep -iE "error" -A6 /tmp/build.log | head -40; echo "BUILD FAILED"; exit 1; }
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-900} -- "$@"
