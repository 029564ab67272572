#!/bin/bash
# SQ counter passes for the headline kernel (c3), each in a run of its own with --pmc only (no trace domains).
# Run on the GPU box:  gpurun -- bash tools/pmc_sq.sh ; summary: gpurun_out/pmc_sq/summary.json (tools/pmc_summary.py)
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
P3="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P GRBM_GUI_ACTIVE --output-format csv -d $O/p$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" > $O/p$i.json 2> $O/p$i.err
  echo "pass $i done"
done
python3 tools/pmc_summary.py $O > $O/summary.json
cat $O/summary.json
