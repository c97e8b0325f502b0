#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/round<N>_* (ROUND, default 5; rounds 3 and 4 were made by the
# same script): for each workload the kernel-trace stats and separate --pmc passes (FETCH_SIZE / WRITE_SIZE / two SQ
# groups; never combined with other trace domains).
#   gpurun --timeout 1200 -- 'bash tools/profile_rounds.sh'      then      python tools/make_profiles_rounds.py
# workloads: head    bench.py's timed loop (Cholesky, N=50 000, 768x512, training iterations)
#            c4      the same at 2040x1356 (BASELINE config 4)
#            batched 24 images per launch (bench.py's `batched` block: tools/batched_bench_scene.py; timings without the
#                    profiler for K = 4 / 8 / 24 on tools/batch_time.py's scenes)
#            c5      rotation-scale model, quantisation-aware iterations, N=30 000 (tools/quant_time.py)
#            fit     24 Kodak images x 10 000 iterations as one batch (tools/kodak_fit.py): trained scenes, prune / grow
#            frozen  bench scene with frozen parameters (tools/static_steps.py): the instruction-count yardstick
#            trained one Kodak picture fitted with the launcher's schedule, then frozen (tools/trained_scene.py)
#            dropin  the drop-in autograd loop under cProfile (tools/profile_autograd_loop.py)
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-6}
OUT=$REPO/gpurun_out/rp$ROUND
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--no-cpu-baseline --images 0 --no-batched --no-static --no-dropin"
C4="$HEAD --height 1356 --width 2040"
stats() { # name, program...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/stats -o run -- python3 "$@" > $OUT/$name.out 2> $OUT/$name.log || true
  rm -f $OUT/$name/stats/*kernel_trace.csv  # the per-dispatch trace is large; the stats are what is kept
}
pmc() { # name, pass, counters (quoted), program...
  local name=$1 pass=$2 ctr=$3; shift 3
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name/$pass -o run -- python3 "$@" > /dev/null 2> $OUT/$name.$pass.log || true
  rm -f $OUT/$name/$pass/*kernel_trace.csv
}
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
mkdir -p $OUT/head $OUT/c4 $OUT/batched $OUT/c5 $OUT/fit $OUT/frozen $OUT/trained
stats head $REPO/bench.py $HEAD
echo "head stats done"
for p in "fetch FETCH_SIZE" "write WRITE_SIZE"; do set -- $p; pmc head $1 "$2" $REPO/bench.py $HEAD --steps 20 --warmup 5; done
pmc head sq1 "$SQ1" $REPO/bench.py $HEAD --steps 20 --warmup 5
pmc head sq2 "$SQ2" $REPO/bench.py $HEAD --steps 20 --warmup 5
echo "head pmc done"
stats c4 $REPO/bench.py $C4
for p in "fetch FETCH_SIZE" "write WRITE_SIZE"; do set -- $p; pmc c4 $1 "$2" $REPO/bench.py $C4 --steps 20 --warmup 5; done
pmc c4 sq1 "$SQ1" $REPO/bench.py $C4 --steps 20 --warmup 5
echo "c4 done"
stats batched $REPO/tools/batched_bench_scene.py 24
python3 $REPO/tools/batch_time.py 50000 512 768 cholesky 4 8 24 > $OUT/batched_plain.out 2>&1 || true
for p in "fetch FETCH_SIZE" "write WRITE_SIZE"; do set -- $p; pmc batched $1 "$2" $REPO/tools/batched_bench_scene.py 24; done
pmc batched sq1 "$SQ1" $REPO/tools/batched_bench_scene.py 24
pmc batched sq2 "$SQ2" $REPO/tools/batched_bench_scene.py 24
for K in 4 8; do
  mkdir -p $OUT/batched$K
  for p in "fetch FETCH_SIZE" "write WRITE_SIZE"; do set -- $p; pmc batched$K $1 "$2" $REPO/tools/batched_bench_scene.py $K; done
done
echo "batched done"
stats c5 $REPO/tools/quant_time.py 30000 400 scale_rot
pmc c5 sq1 "$SQ1" $REPO/tools/quant_time.py 30000 100 scale_rot
echo "c5 done"
stats fit $REPO/tools/kodak_fit.py 24 10000 1
pmc fit sq1 "$SQ1" $REPO/tools/kodak_fit.py 24 1000 1
echo "fit done"
pmc frozen sq1 "$SQ1" $REPO/tools/static_steps.py 30
python3 $REPO/tools/trained_scene.py fit 0 50000 /tmp/trained_scene.pt > $OUT/trained_fit.out 2>&1 || true
pmc trained sq1 "$SQ1" $REPO/tools/trained_scene.py steps 30
stats trained $REPO/tools/trained_scene.py steps 200
echo "frozen + trained scene done"
python3 $REPO/tools/profile_autograd_loop.py 50000 500 > $OUT/dropin_profile.txt 2>&1 || true
python3 $REPO/tools/kodak_fit.py 24 50000 3 > $OUT/kodak50k.out 2>&1 || true
(cd $REPO && python3 tools/lane_model.py > $OUT/lane_model.json 2> /dev/null) || true
echo "dropin + kodak done"
cd $REPO && python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python3 bench.py $C4 > $OUT/c4_plain.json 2> /dev/null          # (no profiler: the c4 figures DESIGN.md quotes)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --images 0 --no-batched --no-static --no-dropin > $OUT/bench_driverlike.json 2> /dev/null
python3 tools/kodak_fit.py 3 50000 3 > $OUT/kodak_shards.out 2>&1 || true
python3 tools/kodak_fit.py 6 50000 3 >> $OUT/kodak_shards.out 2>&1 || true
python3 tools/kodak_fit.py 12 50000 3 >> $OUT/kodak_shards.out 2>&1 || true
tail -n 1 $OUT/bench_plain.json | cut -c1-400
