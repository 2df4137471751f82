"""Async front end: `LLM.generate` streams one `GenerateOutput` per produced token.

Interface of nanovllmx/llm.py:16-131 (same nine constructor arguments, `generate`, `abort`,
`ready`, `shutdown`).  The reference reaches its engine through a spawned process and queues
(core/engine_client.py); here the engine lives in this process and is stepped by one asyncio task,
so a token travels from the sampler to the caller without pickling.
"""
from __future__ import annotations

import asyncio
import dataclasses
import os
import uuid
from typing import AsyncGenerator, Optional

from .core.common import GenerateOutput, SamplingParams
from .core.engine import Engine


class ByteTokenizer:
    """Stand-in used when the model directory holds no tokenizer files (nothing can be downloaded):
    UTF-8 bytes are the token ids.  Enough to drive the engine from strings; not a language tokenizer."""
    eos_token_id = 0

    def encode(self, text: str) -> list:
        return list(text.encode("utf-8"))

    def decode(self, ids: list) -> str:
        return bytes(i & 0xFF for i in ids).decode("utf-8", errors="replace")


def init_tokenizer(model: str):
    has_files = os.path.isdir(model) and any(os.path.exists(os.path.join(model, f))
                                             for f in ("tokenizer.json", "tokenizer_config.json", "vocab.json"))
    if not has_files:
        return ByteTokenizer()
    from transformers import AutoTokenizer

    tok = AutoTokenizer.from_pretrained(model, use_fast=False, trust_remote_code=False)
    if tok.pad_token is None:
        tok.pad_token = tok.eos_token
    return tok


class LLM:
    def __init__(self, model: str, gpu_memory_utilization: float = 0.9, max_bs: int = 50, tp_size: int = 1,
                 pp_size: int = 1, nccl_port: int = 29500, device_ids: Optional[list] = None,
                 enforce_eager: bool = False, context_len: int = 2048, kv_cache_size: Optional[int] = None, seed: int = 0,
                 prefix_reuse: bool = False, max_prefill_tokens: Optional[int] = None, stop_at_context_len: bool = False,
                 chunked_prefill: bool = False, run_ahead: bool = False):
        kwargs = dict(model=model, gpu_memory_utilization=gpu_memory_utilization, max_bs=max_bs, tp_size=tp_size, pp_size=pp_size,
                      nccl_port=nccl_port, device_ids=device_ids, enforce_eager=enforce_eager, context_len=context_len,
                      kv_cache_size=kv_cache_size, seed=seed, prefix_reuse=prefix_reuse, max_prefill_tokens=max_prefill_tokens,
                      stop_at_context_len=stop_at_context_len, chunked_prefill=chunked_prefill, run_ahead=run_ahead)
        self.followers = None
        if tp_size * pp_size > 1:
            # The reference spawns one worker process per GPU here (core/executor.py:32-49).  This engine is SPMD: every
            # rank runs the same loop.  Inside a job that a launcher started (launch.run_ranks, torch.distributed.run) this
            # process is simply its rank; from a plain process the other ranks are started now -- before this process
            # touches the GPU -- and told, event by event, what this process's caller does (launch.EngineFollowers).
            import torch.distributed as dist

            from .launch import EngineFollowers, launcher_env_present, require_launcher
            if not dist.is_initialized() and not launcher_env_present():
                if os.environ.get("NVX_SPAWN_RANKS", "1") == "0":
                    require_launcher(tp_size * pp_size, f"LLM(tp_size={tp_size}, pp_size={pp_size})")
                self.followers = EngineFollowers(tp_size * pp_size, kwargs)
        try:
            self.engine = Engine(**kwargs)
        except BaseException:
            if self.followers is not None:
                self.followers.close(timeout=5.0)
            raise
        self.tokenizer = init_tokenizer(model)
        self.request_states: dict = {}
        self._pump: Optional[asyncio.Task] = None
        self.detokenize_stream = not isinstance(self.tokenizer, ByteTokenizer)

    # -- tokenizer ----------------------------------------------------------------------------------
    def tokenize(self, texts: list) -> list:
        if isinstance(self.tokenizer, ByteTokenizer):
            return [self.tokenizer.encode(t) for t in texts]
        return self.tokenizer(texts)["input_ids"]

    def detokenize(self, token_ids: list) -> list:
        if isinstance(self.tokenizer, ByteTokenizer):
            return [self.tokenizer.decode(t) for t in token_ids]
        return self.tokenizer.batch_decode(token_ids, skip_special_tokens=True)

    # -- engine pump --------------------------------------------------------------------------------
    async def _run_engine(self):
        """Step the engine while it has work.  Every generate() call made before this task first runs
        has already queued its sequence, so a burst of requests is admitted as one prefill batch."""
        eng = self.engine
        try:
            while eng.has_unfinished_sequences() or eng.step_in_flight:
                # waits for the step in flight (or runs one) and queues the next step before this one's tokens
                # are handed out: queue wake-ups, detokenisation and the consumers' own code then run while
                # the device is busy
                if self.followers is not None:
                    self.followers.send({"op": "step"})
                outs = eng.step_overlapped()
                for out in outs:
                    q = self.request_states.get(out.seq_id)
                    if q is None:
                        continue
                    text = self.detokenize([[out.new_token_id]])[0] if self.detokenize_stream else ""
                    if out.is_finished:
                        q.put_nowait(GenerateOutput(token_str=text, is_finished=True, finish_reason=out.finish_reason,
                                                    num_prompt_tokens=out.num_prompt_tokens,
                                                    num_generated_tokens=out.num_generated_tokens,
                                                    token_id=out.new_token_id))
                        q.put_nowait(None)
                        del self.request_states[out.seq_id]
                    else:
                        q.put_nowait(GenerateOutput(token_str=text, is_finished=False, finish_reason=None,
                                                    num_prompt_tokens=0, num_generated_tokens=0, token_id=out.new_token_id))
                await asyncio.sleep(0)   # let consumers drain and new requests arrive
        except Exception as e:   # noqa: BLE001 - a dead pump must not leave its consumers waiting forever
            for q in list(self.request_states.values()):
                q.put_nowait(e)
            self.request_states.clear()
            raise
        finally:
            self._pump = None

    def _ensure_pump(self):
        if self._pump is None or self._pump.done():
            self._pump = asyncio.get_running_loop().create_task(self._run_engine())

    # -- public API ---------------------------------------------------------------------------------
    async def ready(self):
        return True

    async def generate(self, prompts, params: SamplingParams, sequence_id: Optional[str] = None
                       ) -> AsyncGenerator[GenerateOutput, None]:
        if sequence_id is None:
            sequence_id = uuid.uuid4().hex
        try:
            token_ids = list(prompts) if isinstance(prompts, list) else self.tokenize([prompts])[0]
            if params.eos_token_id == -1:
                params.eos_token_id = self.tokenizer.eos_token_id
            q: asyncio.Queue = asyncio.Queue()
            self.request_states[sequence_id] = q
            if self.followers is not None:
                self.followers.send({"op": "add", "id": sequence_id, "ids": token_ids, "params": dataclasses.asdict(params)})
            self.engine.add_sequence(sequence_id=sequence_id, prompt_token_ids=token_ids, sampling_params=params)
            self._ensure_pump()
            while True:
                out = await q.get()
                if out is None:
                    break
                if isinstance(out, BaseException):
                    raise RuntimeError(f"engine step failed: {out}") from out
                yield out
        except asyncio.CancelledError:
            self.abort(sequence_id)
            raise

    def abort(self, sequence_id: str):
        if sequence_id in self.request_states:
            if self.followers is not None:
                self.followers.send({"op": "abort", "id": sequence_id})
            self.engine.abort_sequence(sequence_id)
            del self.request_states[sequence_id]

    def reset(self):
        """Forget every sequence and cached prefix on every rank (between passes of a benchmark)."""
        if self.followers is not None:
            self.followers.send({"op": "reset"})
        self.engine.reset()

    def shutdown(self):
        if self._pump is not None:
            self._pump.cancel()
        if self.followers is not None:
            self.followers.send({"op": "shutdown"})     # the ranks leave the process group together
        self.engine.shutdown()
        if self.followers is not None:
            self.followers.close()
            self.followers = None
