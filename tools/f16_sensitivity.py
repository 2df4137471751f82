#!/usr/bin/env python3
"""How far do float16 logits move when ONE op of the model changes in the last place of some of its outputs?

CPU only.  The float16 oracle (eager rounding points) is run once as it is and once per op with that op's outputs
nudged by a relative 2^-13 before their rounding to float16 -- a quarter of a unit in the last place, which flips the
rounding of about one output in four by one unit and leaves the others bit-identical: the kind of difference two correct
implementations of the op have.  Projections multiply in fp32 on both runs (linear_fp32), same teacher-forced history.
Prints, per op, the largest logit change and its multiple of the 1e-3 * max(1, |logit|) bound."""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle.qwen3_oracle as qo
from nano_vllm_x_amd import qwen3

F16 = torch.float16
hq, hkv, layers = 16, 8, 3
cfg = qwen3.Qwen3Config(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_hidden_layers=layers, num_attention_heads=hq,
                        num_key_value_heads=hkv, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False, torch_dtype="float16")
sd16 = {k: v.to(F16) for k, v in qwen3.random_checkpoint(cfg, 5)}
rng = np.random.default_rng(hq)
prompts = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in [3, 17, 40, 64, 65, 150]]
c = dataclasses.asdict(cfg)

def run(forced=None):
    r = qo.cpu_engine_greedy_oracle(c, sd16, prompts, [10] * 6, 1024, 8, 256, dtype=F16, linear_fp32=True, forced=forced)
    return r, np.stack([torch.stack(r["logits"][i]).double().numpy() for i in range(6)])

base, a = run()
forced = {i: base["streams"][i] for i in range(6)}

def nudged(fn):
    def wrapped(*args, **kw):
        x = args[0]
        args32 = (x.float(),) + tuple(args[1:])
        out = fn(*args32, **kw)      # the op in fp32 on the same float16-valued inputs, not rounded yet
        def fin(t):
            return (t * (1 + 2.0 ** -13)).to(F16)
        return tuple(fin(t) for t in out) if isinstance(out, tuple) else fin(out)
    return wrapped

orig = {n: getattr(qo, n) for n in ("rmsnorm_oracle", "rope_rows_oracle", "silu_mul_oracle", "paged_attention_oracle")}
print(f"{layers} layers, {hq}/{hkv} heads, max|logit| {np.abs(a).max():.2f}")
for name, fn in orig.items():
    setattr(qo, name, nudged(fn))
    _, b = run(forced)
    setattr(qo, name, fn)
    d = np.abs(a - b)
    print(f"{name:26s} last-place nudge: max|dlogit| {d.max():.3e} = {(d / (1e-3 * np.maximum(1, np.abs(a)))).max():.2f} x the 1e-3 bound, rms {np.sqrt((d**2).mean()):.2e}")
