#!/bin/bash
# GPU box: the end-of-round evidence set -> gpurun_out/round/ (copy what is to be judged into profiles/ afterwards).
# usage: bash tools/round_end.sh <tag>     e.g. r01h
set -o pipefail
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/round
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $out/$name.log; return 1; }
        grep "^{" $out/$name.log > $out/${tag}_$name.json; cut -c1-160 $out/${tag}_$name.json; }
run bench_3b_default || exit 1
run bench_3b_route_reference --route reference --no-cpu-baseline || exit 1
run bench_3b_hybrid --scenario hybrid --no-cpu-baseline || exit 1
run bench_3b_mxfp8 --scenario mxfp8 --no-cpu-baseline || exit 1
run bench_1b_default --model llama-3.2-1b --no-cpu-baseline || exit 1
run bench_8b_hybrid_b12 --model llama-3.1-8b --batch 12 --scenario hybrid --no-cpu-baseline || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $out/under_rocprof.log 2>&1 || { echo "rocprof run FAILED"; tail -5 $out/under_rocprof.log; exit 1; }
grep "^{" $out/under_rocprof.log > $out/${tag}_bench_under_rocprof.json
cp $out/prof/*/*_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
# the --stats table sums the whole process (initialisation, warm-up, the clock probe's and the grouped-GEMM autotune's launches): cut the steady steps out of the trace
python3 $GRAFT_REPO_ROOT/tools/trace_steady_stats.py $out/prof/*/*_kernel_trace.csv $out/${tag}_bench_kernel_stats_steady.csv 4
echo done
