#!/bin/bash
# Run on the GPU box: isolated per-kernel durations (rocprofv3 --kernel-trace, synchronous steps) of the default library at
# several frames-per-stream counts (64 streams): how the kernels scale with the number of rounds a launch makes.
# usage: tools/ab_sizes.sh 16 32 64 128 256
for F in "$@"; do
  echo "frames per stream $F"
  tools/ab_kernels.sh atracdenc_amd/libat3hip.so -- --frames $F | head -1
done
