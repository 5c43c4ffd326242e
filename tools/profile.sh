#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: kernel-trace stats, then PMC counters in separate passes
# (never combined with sys/hip tracing).  Usage on the GPU box: bash tools/profile.sh <tag> <config> [bench args]
# MULTI-RANK profiles (an 8-GPU node): wrap EACH RANK's python directly and export the rendezvous by hand — never the
# self-launching parent (`bench.py --gpus N` without RANK / WORLD_SIZE starts torch.distributed.run as a child; rocprofv3's
# preloaded library would initialise the GPU in the parent, and the launcher's re-exec is then the forbidden exec):
#   for r in 0 1 2 3 4 5 6 7; do
#     RANK=$r LOCAL_RANK=$r WORLD_SIZE=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 \
#       rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rank$r -- python3 bench.py --gpus 8 --config c5 &
#   done; wait
# (bench.py sees RANK / WORLD_SIZE and joins the group instead of launching; the JSON line of rank 0 carries n_ranks_seen,
#  ms_per_step_per_rank {min, max, all} and the all-reduce route.)
set -u
TAG=${1:-r02}; CFG=${2:-c2}; shift; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_$CFG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-sustained $*"
# the kernel-trace pass runs bench.py with its DEFAULT step / warm-up counts (what the driver runs), so that the
# average duration in kernel_stats.csv is the average bench.py itself reports (only the CPU baseline leg is skipped)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --config $CFG --no-cpu-baseline $* > $OUT/trace_bench.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_lds.log
cd $REPO
MIMO_PROFILE_TAG=$TAG python3 tools/summarize_profile.py $OUT $CFG > $OUT/summary.txt 2>&1
tail -1 $OUT/trace_bench.json > $OUT/bench_line.json
echo "--- $CFG"; head -8 $OUT/summary.txt; tail -2 $OUT/summary.txt
