#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_bwt.sh r02_bwt'): rocprofv3 passes over the block sort
# (tools/bwt_rate.py: forward + inverse of 1 GiB per workload).  Kernel trace + stats in one pass, each PMC group in its
# own pass (every --pmc pass carries --kernel-trace and nothing else: no sys / runtime / hip / hsa / memory-copy / marker trace next to counters).  Then
#   python profiles/summarize.py <tag> gpurun_out/<tag>_stats --side --config workload=canterbury,bytes=1073741824,block=32768 --pmc fetch=... write=... sq1=... sq2=...
set -e -o pipefail
TAG=${1:-r02_bwt}
WL=${2:-canterbury}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
RUN="python3 $ROOT/tools/bwt_rate.py --workloads $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- $RUN --out "$OUT/${TAG}_rate_under_rocprof.jsonl" > /dev/null 2> "$OUT/${TAG}_stats.log"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_fetch" -- $RUN --out /dev/null > /dev/null 2> "$OUT/${TAG}_pmc_fetch.log"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_write" -- $RUN --out /dev/null > /dev/null 2> "$OUT/${TAG}_pmc_write.log"
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_sq1" -- $RUN --out /dev/null > /dev/null 2> "$OUT/${TAG}_pmc_sq1.log"
echo "sq1 done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_sq2" -- $RUN --out /dev/null > /dev/null 2> "$OUT/${TAG}_pmc_sq2.log"
echo "sq2 done"
find "$OUT" -path "*${TAG}_*" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" ! -name "*.jsonl" -delete 2>/dev/null || true
du -sh "$OUT"/${TAG}_* | tail -8
