#!/bin/bash
# Fast experimental build of the Mode A kernels only: compiles gi_dense.hip and gi_dense_host.cpp with extra flags (e.g.
# -DJRLQP_ROT_PROF: the sweep's time line) through the same device pipeline as the Makefile and links them with the objects of the
# last full build into tools/ubench/lib$1.bin.  On the GPU box: cp tools/ubench/libP.bin jrl-qp_amd/csrc/libjrlqp_hip.so first.
#   usage: tools/ubench/exp_dense.sh [TAG] [extra hipcc flags]
set -e
TAG=${1:-P}; shift || true
ROOT=$(cd $(dirname $0)/../.. && pwd); C=$ROOT/jrl-qp_amd/csrc; W=/tmp/expd_$TAG; mkdir -p $W
L=/opt/rocm/lib/llvm/bin
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-command-line-argument -I$C $*"
/opt/rocm/bin/hipcc $F --offload-device-only -S $C/gi_dense.hip -o $W/d.raw.s
python3 $ROOT/tools/isa_lint.py --fix $W/d.raw.s $W/d.s
$L/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/d.s -o $W/d.dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/d.co $W/d.dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/d.co -output=$W/d.hipfb
/opt/rocm/bin/hipcc $F --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/d.hipfb -c $C/gi_dense.hip -o $W/gi_dense.o
/opt/rocm/bin/hipcc $F -x hip -c $C/gi_dense_host.cpp -o $W/gi_dense_host.o
OBJS=""; for o in gi_batch gi_batch_wide gi_batch_wide2 gi_batch_big structured blockgi selftest capi; do OBJS="$OBJS $C/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ubench/lib$TAG.bin $W/gi_dense.o $W/gi_dense_host.o $OBJS
ls -la $ROOT/tools/ubench/lib$TAG.bin
