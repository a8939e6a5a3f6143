#!/bin/bash
# Experiment build of libconvexadam_hip.so with -DCVX_BM_PHASES (per-phase shader clocks inside the steps of the marching three-box kernels)
# -> convexadam_amd/csrc/libconvexadam_hip_phases.so ; run tools/box_phases.py on the GPU box with CONVEXADAM_HIP_LIB pointing at it.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/convexadam_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -fvisibility=hidden -Wno-unused-function -I$R/include -I$S -DCVX_BUILDING=1"
mkdir -p $S/build_phases
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize -DCVX_BM_PHASES=1 -c $S/boxmarch.hip -o $S/build_phases/boxmarch.o 2>&1 | grep -i "error" || true
OBJS=$(ls $S/build/*.o | grep -v boxmarch.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $S/libconvexadam_hip_phases.so $OBJS $S/build_phases/boxmarch.o
echo $S/libconvexadam_hip_phases.so
