#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel of one source file (device-only compile, metadata notes of the code object):
#   tools/kernel_resources.sh boxmarch.hip [filter]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1; shift || true
EXTRA=""
case $F in warp.hip|adamfast.hip|boxmarch.hip|boxtile.hip|corrbox.hip|corrfused.hip|mindmarch.hip) EXTRA="-fno-slp-vectorize";; esac
O=/tmp/kres_$$.co
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -I$R/include -I$R/convexadam_amd/csrc \
  $EXTRA -DCVX_BUILDING=1 --cuda-device-only -c $R/convexadam_amd/csrc/$F -o $O
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$O --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$O.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $O.elf | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if flt in name: print('%-110s vgpr %3s sgpr %3s lds %6s scratch %4s wg %s'%(name[:110],g('vgpr_count'),g('sgpr_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size'),g('max_flat_workgroup_size')))
" "$@"
rm -f $O $O.elf
