#!/bin/bash
# All rocprofv3 runs behind profiles/<tag>_*: run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh').
#   kernel trace of `python bench.py` (configs[1] headline) and of queued generation at configs[3] (tools/prof_gen128.py), and for configs[1] (tools/prof_train.py) and configs[3]
#   (tools/prof_cfg4.py): kernel trace + SEPARATE --pmc passes for FETCH_SIZE, WRITE_SIZE and the SQ counters
#   (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no --pmc together with trace domains other than
#   --kernel-trace).  Outputs under gpurun_out/prof_<tag>/; tools/summarise_profiles.py turns them into
#   gpurun_out/prof_<tag>_summary/<tag>_*.csv, which are then copied into profiles/.   usage: collect_profiles.sh [tag]
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, program...
  local name=$1; shift
  echo "== $name"; timeout -k 10 280 rocprofv3 --output-format csv "${@:1:$#}" || { echo "FAILED $name"; exit 1; }
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
run bench_ks --kernel-trace --stats -d $OUT/bench_ks -o bench -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-generation --no-other-configs --no-strong-leg > $OUT/bench_ks.log 2>&1
for cfg in train cfg4; do
  prog=$ROOT/tools/prof_${cfg}.py
  run ${cfg}_ks --kernel-trace --stats -d $OUT/${cfg}_ks -o $cfg -- python3 $prog > $OUT/${cfg}_ks.log 2>&1
  run ${cfg}_f --kernel-trace --pmc FETCH_SIZE -d $OUT/${cfg}_f -o $cfg -- python3 $prog > $OUT/${cfg}_f.log 2>&1
  run ${cfg}_w --kernel-trace --pmc WRITE_SIZE -d $OUT/${cfg}_w -o $cfg -- python3 $prog > $OUT/${cfg}_w.log 2>&1
  run ${cfg}_sq --kernel-trace --pmc $SQ -d $OUT/${cfg}_sq -o $cfg -- python3 $prog > $OUT/${cfg}_sq.log 2>&1
done
run gen128_ks --kernel-trace --stats -d $OUT/gen128_ks -o gen128 -- python3 $ROOT/tools/prof_gen128.py > $OUT/gen128_ks.log 2>&1
# summaries only travel back (the raw databases and counter tables are tens of MB)
# (the raw outputs are deleted only after the summariser succeeded: a failed summary must not cost a second profiling run)
python3 $ROOT/tools/summarise_profiles.py $TAG $ROOT/gpurun_out/prof_${TAG}_summary > $OUT/../prof_${TAG}_summary.log 2>&1 \
  || { echo "FAILED summarise (raw outputs kept in $OUT)"; tail -20 $OUT/../prof_${TAG}_summary.log; exit 1; }
rm -rf $OUT
ls -la $ROOT/gpurun_out/prof_${TAG}_summary
