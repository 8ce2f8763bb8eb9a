#!/bin/bash
# Run on the GPU box: tools/at1_bench.py with atracdenc_amd/libat3hip_prev.so (a build of an older commit, made by hand) and
# with atracdenc_amd/libat3hip.so, alternating on the same box.
for i in 1 2 3; do for L in libat3hip_prev.so libat3hip.so; do
AT3HIP_LIB=$PWD/atracdenc_amd/$L python tools/at1_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', d.get('value'), d.get('ms_per_step'), d.get('device_ms'))"
done; done
