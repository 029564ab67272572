#!/bin/bash
# Register / spill / LDS figures of the kernels in a built object (code-object metadata), filtered by a name pattern:
#   tools/kernel_regs.sh [pattern] [object]      default: the c3 instantiation of gi_batch_kernel in gi_batch_wide.o
P="${1:-gi_batch_kernelILi50ELi2ELb0ELb0ELb0ELb0}"
OBJ="${2:-$(dirname $0)/../jrl-qp_amd/csrc/gi_batch_wide.o}"
O=/tmp/jrlqp_co_$$; mkdir -p $O
objcopy -O binary --only-section=.hip_fatbin "$OBJ" $O/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$O/fat.bin --output=$O/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $O/k.co | awk -v p="$P" '/\.name:/{name=$2} /\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|group_segment_fixed|\.sgpr_count/{ if(name ~ p) print name, $1, $2}' | sed 's/_ZN5jrlqp12_GLOBAL__N_1//' | sort | uniq
rm -rf $O
