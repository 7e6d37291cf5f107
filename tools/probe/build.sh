#!/bin/bash
# builds the probe kernels (tools/bench_starve.py); the .so is git-ignored and travels with the gpurun snapshot
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared probe_kernels.hip -o libprobe.so
