"""GPU box: the reference's multi-GPU wrappers (FSDP FULL_SHARD per decoder layer, DDP) around the FP8 model, launched
exactly like the driver launches bench.py (torch.distributed.run, nccl = RCCL), at world size 1 -- the single GPU of
the box.  Exercises flat-parameter views, hooks, clip_grad_norm_ and the amax arenas under the wrappers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode,scenario", [("fsdp_full", "default"), ("ddp", "default"), ("fsdp_full", "mxfp8")])
def test_train_harness_under_wrappers_world1(dev, mode, scenario):
    env = dict(os.environ, LLM_FP8_AMD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", "-m", "llm_fp8_amd.train", "--model_name", "llama-3.2-1b", "--num_hidden_layers", "2",
           "--vocab_size", "4096", "--batch_size", "4", "--max_seq_length", "128", "--mixed_precision", "fp8", "--use_te",
           "--fp8_scenario", scenario, "--sharding_mode", mode, "--num_steps", "4", "--learning_rate", "1e-3",
           "--num_warmup_steps", "0", "--repeat_batch"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 4 and all(l["loss"] == l["loss"] for l in lines), r.stdout[-2000:]
    assert lines[-1]["loss"] < lines[0]["loss"]
