"""Formula-defined weights of the tiny Qwen3 used by the golden logits (no RNG dependence)."""
import torch


def formula_weight(shape, salt: int, scale: float) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    w = torch.sin(i * 0.61803398875 + salt * 1.37) * torch.cos(i * 0.0137 + salt)
    return (w * scale).reshape(shape).to(torch.bfloat16)


def tiny_state_dict(c):
    h, d, hq, hkv, it = (c["hidden_size"], c["head_dim"], c["num_attention_heads"],
                         c["num_key_value_heads"], c["intermediate_size"])
    sd = {"model.embed_tokens.weight": formula_weight((c["vocab_size"], h), 1, 1.0),
          "model.norm.weight": formula_weight((h,), 2, 0.2) + 1.0,
          "lm_head.weight": formula_weight((c["vocab_size"], h), 3, 0.3)}
    for l in range(c["num_hidden_layers"]):
        p = f"model.layers.{l}."
        s = 10 * (l + 1)
        sd[p + "self_attn.q_proj.weight"] = formula_weight((hq * d, h), s + 1, 0.15)
        sd[p + "self_attn.k_proj.weight"] = formula_weight((hkv * d, h), s + 2, 0.15)
        sd[p + "self_attn.v_proj.weight"] = formula_weight((hkv * d, h), s + 3, 0.15)
        sd[p + "self_attn.o_proj.weight"] = formula_weight((h, hq * d), s + 4, 0.1)
        sd[p + "self_attn.q_norm.weight"] = formula_weight((d,), s + 5, 0.2) + 1.0
        sd[p + "self_attn.k_norm.weight"] = formula_weight((d,), s + 6, 0.2) + 1.0
        sd[p + "mlp.gate_proj.weight"] = formula_weight((it, h), s + 7, 0.15)
        sd[p + "mlp.up_proj.weight"] = formula_weight((it, h), s + 8, 0.15)
        sd[p + "mlp.down_proj.weight"] = formula_weight((h, it), s + 9, 0.1)
        sd[p + "input_layernorm.weight"] = formula_weight((h,), s + 10, 0.2) + 1.0
        sd[p + "post_attention_layernorm.weight"] = formula_weight((h,), s + 11, 0.2) + 1.0
    return {k: v.to(torch.bfloat16) for k, v in sd.items()}
