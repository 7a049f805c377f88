#!/bin/bash
# Collects the profiles committed under profiles/: kernel stats, FETCH_SIZE / WRITE_SIZE and SQ
# counter passes of the headline bench (each counter pass in a run of its own), then condenses them.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r03
set -e
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
python3 tools/summarize_profile.py $TAG $OUT/stats $OUT/fetch $OUT/write $OUT/sq > $OUT/summary.log 2>&1
grep '^{' $OUT/bench_under_rocprof.log | tail -1 > profiles/${TAG}_bench_under_rocprof.log
cp profiles/${TAG}_* $OUT/
