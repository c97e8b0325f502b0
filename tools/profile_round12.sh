#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/: kernel-trace stats of the default bench.py run and four
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / two SQ groups).  Run on the GPU box:
#   gpurun -- 'bash tools/profile_round12.sh'      then      python tools/make_profiles12.py round2
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/rp
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/bench.py --no-cpu-baseline --images 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
PMC_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --images 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o run -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o run -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/write.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq1 -o run -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/sq1.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o run -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/sq2.log
cd $REPO && python3 bench.py --train-step --images-per-gpu-probe > $OUT/bench_plain.json 2> $OUT/bench_plain.err
tail -n 1 $OUT/bench_plain.json | cut -c1-600
