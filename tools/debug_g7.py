"""Debug: 3-step loss curves of the G7 tiny model under several precisions / wrappers (GPU box)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers.models.llama.modeling_llama import LlamaForCausalLM
from llm_fp8_amd import llama, train
from llm_fp8_amd.pytorch.fp8 import FP8GlobalStateManager

meta = json.load(open("tests/golden/meta.json"))["g7"]
dev = torch.device("cuda:0")
config = llama.llama_config("llama-3.2-1b", num_hidden_layers=2, vocab_size=4096)


def hf_model():
    torch.manual_seed(42)
    prev = torch.get_default_dtype(); torch.set_default_dtype(torch.bfloat16)
    try:
        return LlamaForCausalLM(config)
    finally:
        torch.set_default_dtype(prev)


def run(model, steps=4):
    model.train()
    ids = torch.tensor(meta["input_ids"], device=dev)
    batch = {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids.clone()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    out = []
    for _ in range(steps):
        o = model(**batch); o.loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step(); opt.zero_grad(); out.append((round(float(o.loss.detach()), 3), round(float(gn), 2)))
    return out

print("golden", meta["loss"], meta["grad_norm"])
hf = hf_model()
print("tied:", hf.lm_head.weight.data_ptr() == hf.model.embed_tokens.weight.data_ptr(), config.tie_word_embeddings)
print("hf bf16 gpu     ", run(hf_model().to(dev)))
for mp, use_te, scen in (("bf16", True, "default"), ("fp8", True, "default"), ("fp8", True, "mxfp8"), ("fp8", False, "default")):
    FP8GlobalStateManager.reset()
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=1, max_seq_length=128, mixed_precision=mp, use_te=use_te,
                               fp8_scenario=scen, num_hidden_layers=2, vocab_size=4096, num_warmup_steps=0, learning_rate=1e-3)
    m = hf_model()
    if use_te:
        m = llama.TELlamaForCausalLM.from_hf_state_dict(m.state_dict(), config, scen)
    m = train.prepare_model(m.to(dev), cfg)
    print(f"{mp} te={use_te} {scen}", run(m), "tied:", m.lm_head.weight.data_ptr() == m.model.embed_tokens.weight.data_ptr())
