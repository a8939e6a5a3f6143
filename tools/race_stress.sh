#!/bin/bash
# Race stress of the specialised-wavefront pipelines (run on the GPU box through gpurun, after `python -m convexadam_amd.csrc.build --jitter`
# in the build container): the bit-exact GPU tests are repeated against libconvexadam_hip_jitter.so, in which every wavefront sleeps a
# pseudo-random time on both sides of every workgroup barrier.  A missing barrier or a ring slot that is reused too early turns into a
# mismatch against the oracle.  Output: gpurun_out/race_stress.txt (copy to profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export CONVEXADAM_HIP_LIB=$R/convexadam_amd/csrc/libconvexadam_hip_jitter.so
O=$R/gpurun_out/race_stress.txt
echo "library: $CONVEXADAM_HIP_LIB (CVX_RACE_JITTER build); 3 repetitions of the bit-exact operator / pipeline tests" > $O
for rep in 1 2 3; do
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_modes.py tests/test_gpu_surfdist.py -q -x -p no:cacheprovider \
    -k "vs_oracle or bit_exact or bit_identical or variants or marching or worst_case or snapshots or kernel_variants or x_tiles or torch_mean or reference or fp16 or workgroups_per_cu or search_widths or nnunet or threads or fast or hd95 or edt or drop_in or surface or label_bits" 2>&1 | tail -3 >> $O
  timeout 900 python -m pytest tests/test_gpu_mind_single.py tests/test_gpu_certified.py tests/test_gpu_ic_fused.py -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
done
python - >> $O <<'PY'
import ctypes, os
L = ctypes.CDLL(os.environ["CONVEXADAM_HIP_LIB"]); print("loaded jitter library, version", L.cvx_version())
PY
