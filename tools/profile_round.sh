#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r01
# 1. --kernel-trace --stats of the default bench command   -> kernel durations
# 2. --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes  -> HBM-side traffic per kernel
# 3. --pmc SQ_* on the same command                        -> issue / wait picture of the elimination kernel
# 4. --kernel-trace --stats of bench.py --workload c5      -> kernel durations of the batched workload
# Raw outputs land in gpurun_out/prof_<round>/ ; tools/profile_summarize.py condenses them into profiles/.
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq -o s --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/trace_c5 -o t --output-format csv -- python3 $ROOT/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_c5.log 2>&1 || exit 1
# 5. o_v = random at the same size: the multi-CU dataflow kernel (rlap_flow.hip) -- durations, waves / busy CUs, HBM-side traffic
rocprofv3 --kernel-trace --stats -d $OUT/trace_rand -o t --output-format csv -- python3 $ROOT/bench.py --o_v random --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace_rand.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/sq_rand -o s --output-format csv -- python3 $ROOT/bench.py --o_v random --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq_rand.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_rand -o f --output-format csv -- python3 $ROOT/bench.py --o_v random --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch_rand.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_rand -o w --output-format csv -- python3 $ROOT/bench.py --o_v random --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write_rand.log 2>&1 || exit 1
cd $ROOT && python3 tools/profile_summarize.py $R $OUT > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
