#!/bin/bash
# does the library's own default for GPU_MAX_HW_QUEUES take effect when it is loaded before any HIP call?
cd $GRAFT_REPO_ROOT
echo "== env unset (library constructor sets 16)"; env -u GPU_MAX_HW_QUEUES timeout -k 10 200 python tools/quick_streams.py 4096 16 2>&1 | grep streams
echo "== env 4"; GPU_MAX_HW_QUEUES=4 timeout -k 10 200 python tools/quick_streams.py 4096 16 2>&1 | grep streams
