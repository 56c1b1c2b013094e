#!/bin/bash
# Why is the one-state rANS encoder slower with fewer blocks per wave?    The same buffer (1 GiB of
# Zipf bytes, 256 KiB blocks = 4096 blocks) with 16 and with 4 blocks per wave, under the SQ counter groups of
# tools/profile_round.sh.  Run on the GPU box; output gpurun_out/r03_rans1_lanes_<lanes>_<group>/.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp
export TMPDIR=/tmp
for LANES in 16 4; do
  export RCX_RANS1_LANES=$LANES
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/r03_rans1_lanes_${LANES}_sq1" -- python3 $ROOT/tools/sweep.py --coder 2 --blocks 262144 --workloads zipf --out $OUT/r03_tmp.jsonl > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/r03_rans1_lanes_${LANES}_sq2" -- python3 $ROOT/tools/sweep.py --coder 2 --blocks 262144 --workloads zipf --out $OUT/r03_tmp.jsonl > /dev/null 2>&1
  rocprofv3 --pmc SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d "$OUT/r03_rans1_lanes_${LANES}_sq3" -- python3 $ROOT/tools/sweep.py --coder 2 --blocks 262144 --workloads zipf --out $OUT/r03_tmp.jsonl > /dev/null 2>&1 || echo "group 3 not available"
  echo "lanes $LANES done"
done
find "$OUT" -path "*r03_rans1_lanes_*" -type f ! -name "*counter_collection.csv" -delete 2>/dev/null || true
