out=gpurun_out/r3s; mkdir -p $out
run() { name=$1; shift; env HSA_ENABLE_IPC_MODE_LEGACY=0 LLM_FP8_AMD_FORCE_DIST=1 LLM_FP8_AMD_FORCE_COLLECTIVES=1 "$@" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29539 tests/fsdp_fp8_worker.py $SCEN > $out/$name.log 2>&1; echo "$name: $(grep '^{' $out/$name.log | cut -c1-330)"; }
SCEN=default run nobias LLM_FP8_AMD_NO_MLP_BIAS_FUSION=1
SCEN=mxfp8 run mx LLM_FP8_AMD_X=1
SCEN=default run nohandoff LLM_FP8_AMD_NO_DY_HANDOFF=1
SCEN=default run nogroup LLM_FP8_AMD_GROUPED_GEMM=off
