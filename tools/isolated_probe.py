#!/usr/bin/env python3
"""Where the float16 end-to-end logit error of the hand-written path comes from (projections in fp32 on both sides):
HIP fused path, HIP op-by-op path, the oracle in float16 (eager rounding points) and the oracle in fp32, pairwise."""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NVX_FP32_LINEAR"] = "1"
import numpy as np, torch
from nano_vllm_x_amd import qwen3
from nano_vllm_x_amd.core.engine import EngineCore
from nano_vllm_x_amd.core.model_runner import ModelRunner
from oracle.qwen3_oracle import cpu_engine_greedy_oracle
from test_gpu_configs import _forced_run

F16 = torch.float16
hq, hkv = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 8
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = qwen3.Qwen3Config(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_hidden_layers=layers, num_attention_heads=hq,
                        num_key_value_heads=hkv, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False, torch_dtype="float16")
sd16 = {k: v.to(F16) for k, v in qwen3.random_checkpoint(cfg, 5)}
rng = np.random.default_rng(hq)
lens = [3, 17, 40, 64, 65, 150]
prompts = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
n_new, slots, max_bs, ctx = 10, 1024, 8, 256
c = dataclasses.asdict(cfg)
ref16 = cpu_engine_greedy_oracle(c, sd16, prompts, [n_new] * 6, slots, max_bs, ctx, dtype=F16, linear_fp32=True)
forced = {i: ref16["streams"][i] for i in range(6)}
ref32 = cpu_engine_greedy_oracle(c, {k: v.float() for k, v in sd16.items()}, prompts, [n_new] * 6, slots, max_bs, ctx, forced=forced)

def stack(r):
    return np.stack([torch.stack(r["logits"][i]).double().numpy() for i in range(6)])

def hip(fused, eager):
    runner = ModelRunner(model="isolated", max_bs=max_bs, enforce_eager=eager, context_len=ctx, config=cfg)
    runner.load_model(state_dict_iter=sd16.items())
    runner.model.fused = fused
    runner.initialize_kv_cache(slots)
    core = EngineCore(kv_cache_size=slots, max_bs=max_bs, context_len=ctx)
    with torch.inference_mode():
        _t, lg, _ = _forced_run(core, runner, [(str(i), p, n_new) for i, p in enumerate(prompts)], {str(i): forced[i] for i in range(6)})
    return np.stack([np.stack(lg[str(i)]).astype(np.float64) for i in range(6)])

def gap(a, b, name):
    d = np.abs(a - b)
    r = d / (1e-3 * np.maximum(1.0, np.abs(b)))
    print(f"{name:58s} max|d| {d.max():.3e}  prefill-step max {d[:, 0].max():.3e}  x(1e-3 bound) {r.max():.2f}  rms {np.sqrt((d**2).mean()):.2e}")

a16, a32 = stack(ref16), stack(ref32)
print(f"heads {hq}/{hkv}, {layers} layers, max|logit| {np.abs(a32).max():.2f}")
gap(a16, a32, "oracle float16 (eager rounding) vs oracle fp32")
for fused, eager in ((True, False), (False, True), (True, True)):
    g = hip(fused, eager)
    tag = f"HIP {'fused' if fused else 'op-by-op'} {'eager' if eager else 'graph'}"
    gap(g, a16, tag + " vs oracle float16")
    gap(g, a32, tag + " vs oracle fp32")
