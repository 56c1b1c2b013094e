#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r02'): the rocprofv3 passes bench.py's roofline block and
# profiles/ are built from.  Kernel trace + stats in one pass, each PMC group in its own pass (FETCH_SIZE and WRITE_SIZE
# do not fit the TCC slots together; every --pmc pass carries --kernel-trace and nothing else: no sys / runtime / hip / hsa / memory-copy / marker trace next to counters).  Outputs under gpurun_out/<tag>_*;
# `python profiles/summarize.py <tag> gpurun_out/<tag>_stats --pmc ...` then turns them into profiles/<tag>_*.
set -e -o pipefail
TAG=${1:-r03}
shift || true   # anything after the tag goes to bench.py (e.g. --coder rans8 --workload zipf: the sibling coders, summarize.py --side)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-end-to-end $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- $BENCH --steps 5 --warmup 2 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_stats.log"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_fetch" -- $BENCH --steps 2 --warmup 1 > /dev/null 2> "$OUT/${TAG}_pmc_fetch.log"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_write" -- $BENCH --steps 2 --warmup 1 > /dev/null 2> "$OUT/${TAG}_pmc_write.log"
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_sq1" -- $BENCH --steps 2 --warmup 1 > /dev/null 2> "$OUT/${TAG}_pmc_sq1.log"
echo "sq1 done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_sq2" -- $BENCH --steps 2 --warmup 1 > /dev/null 2> "$OUT/${TAG}_pmc_sq2.log"
echo "sq2 done"
# keep only the CSVs (the merge back is limited to 64 MiB)
find "$OUT" -path "*${TAG}_*" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" -delete 2>/dev/null || true
du -sh "$OUT"/${TAG}_* | tail -8
