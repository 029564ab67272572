#!/bin/bash
# Fast experimental build of the headline kernel only: compiles gi_batch_wide.hip with -DJRLQP_EXP_ONLY_C3 (just the cold n = 50
# instantiations, production + phase profile) through the same device pipeline as the Makefile (hipcc -S, isa_lint --fix, assemble,
# bundle, host compile) and links it with the objects of the last full build into tools/ubench/lib$1.bin (default B).
# tools/ubench/ab.sh then times libA.bin against libB.bin on the GPU box.   usage: tools/ubench/exp_c3.sh [A|B] [extra hipcc flags]
set -e
TAG=${1:-B}; shift || true
ROOT=$(cd $(dirname $0)/../.. && pwd); C=$ROOT/jrl-qp_amd/csrc; W=/tmp/exp_$TAG; mkdir -p $W
L=/opt/rocm/lib/llvm/bin
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-command-line-argument -DJRLQP_EXP_ONLY_C3 -I$C $*"
/opt/rocm/bin/hipcc $F --offload-device-only -S $C/gi_batch_wide.hip -o $W/wide.raw.s
python3 $ROOT/tools/isa_lint.py --fix $W/wide.raw.s $W/wide.s
$L/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/wide.s -o $W/wide.dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/wide.co $W/wide.dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/wide.co -output=$W/wide.hipfb
/opt/rocm/bin/hipcc $F --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/wide.hipfb -c $C/gi_batch_wide.hip -o $W/gi_batch_wide.o
OBJS=""; for o in gi_batch gi_batch_wide2 gi_batch_big gi_dense gi_dense_host structured blockgi blockgi_c9 selftest capi; do OBJS="$OBJS $C/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ubench/lib$TAG.bin $W/gi_batch_wide.o $OBJS
grep -c "scratch_" $W/wide.s | sed "s/^/scratch ops in listing: /"
ls -la $ROOT/tools/ubench/lib$TAG.bin
