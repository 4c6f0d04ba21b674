#!/bin/bash
# GPU box: the PMC evidence passes (each counter set in its OWN rocprofv3 run, --kernel-trace only) -> gpurun_out/pmc/<tag>_*
# usage: bash tools/pmc_round.sh <tag>
set -o pipefail
tag=${1:-rXX}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; counters=$2; script=$3
  timeout -k 10 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out/$name -- python3 $root/tools/$script 3 > $out/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $out/$name.log; return 1; }
  cp $out/$name/*/*_counter_collection.csv $out/${tag}_$name.csv; echo "$name ok"; }
# the GEMM launch mix of the 3B step (forward sites, grouped backward pairs, lm_head): tools/pmc_gemm_step.py writes its manifest
passm() { name=$1; counters=$2
  timeout -k 10 400 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out/$name -- python3 $root/tools/pmc_gemm_step.py 3 $out/${tag}_gemm_manifest.json > $out/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $out/$name.log; return 1; }
  cp $out/$name/*/*_counter_collection.csv $out/${tag}_$name.csv; echo "$name ok"; }
passm gemm_fetch FETCH_SIZE || exit 1
passm gemm_write WRITE_SIZE || exit 1
passm gemm_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES" || exit 1
pass hbm_fetch FETCH_SIZE pmc_hbm_kernels.py || exit 1
pass hbm_write WRITE_SIZE pmc_hbm_kernels.py || exit 1
echo done
