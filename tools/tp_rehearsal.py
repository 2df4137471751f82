#!/usr/bin/env python3
"""Rehearsal of the multi-rank engine path on a one-GPU box: N ranks share cuda:0 and talk through
gloo (NVX_TP_BACKEND=gloo), a small Qwen3 geometry, the LLM.generate pump.

  NVX_TP_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29513 tools/tp_rehearsal.py
  NVX_PP=2 ... --nproc-per-node 4 ...     # two stages of two tensor-parallel ranks

bench.py runs the same way (NVX_TP_BACKEND=gloo python -m torch.distributed.run ... bench.py --gpus N --num-seqs 48
--max-bs 128 --kv-cache-size 200000 --no-roofline --no-cpu-baseline).  With ranks SHARING a GPU and decode batches of
112 rows or more, also set NVX_LIN_GATEUP_WIDE_MIN=100000: the 64-row gate/up projection takes whole SIMDs' registers and
cannot be placed while the other ranks' all-reduce kernels wait on every CU for this rank (on separate GPUs they never
share a chip).
"""
import asyncio, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nano_vllm_x_amd import LLM, SamplingParams
from nano_vllm_x_amd.qwen3 import PRESETS

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
PRESETS["rehearsal"] = dict(vocab_size=2048, hidden_size=512, intermediate_size=1536, num_hidden_layers=3,
                            num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=512, tie_word_embeddings=False)


def log(msg):
    print(f"[rank {rank} {time.strftime('%H:%M:%S')}] {msg}", flush=True)


async def main(llm):
    import random
    rnd = random.Random(0)
    reqs = [([rnd.randint(0, 2000) for _ in range(rnd.randint(5, 40))], rnd.randint(3, 12)) for _ in range(6)]

    async def one(i, p, n):
        toks = []
        async for o in llm.generate(list(p), SamplingParams(temperature=0.6, ignore_eos=True, max_new_tokens=n), sequence_id=f"r{i}"):
            toks.append(o.token_id)
        return toks

    return await asyncio.gather(*(one(i, p, n) for i, (p, n) in enumerate(reqs)))


import faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("NVX_REHEARSAL_DUMP_S", "60")), exit=True)   # where is a stuck rank?
log("building")
pp = int(os.environ.get("NVX_PP", "1"))     # pipeline stages; the remaining factor of the world size is tensor parallel
llm = LLM(model="rehearsal", max_bs=8, tp_size=world // pp, pp_size=pp, context_len=128, kv_cache_size=2048, enforce_eager=True)
log("built")
res = asyncio.run(main(llm))
digest = hashlib.sha256(repr(res).encode()).hexdigest()[:16]
log(f"generated {sum(len(r) for r in res)} tokens, digest {digest}, steps {llm.engine.model_runner.stats}")
if world > 1:
    t = torch.tensor([int(digest, 16) % (2 ** 62)], dtype=torch.int64, device="cuda")
    ref = t.clone()
    torch.distributed.broadcast(ref, src=0)
    assert torch.equal(t, ref), "ranks generated different tokens"
    log("ranks agree")
llm.shutdown()
faulthandler.cancel_dump_traceback_later()
