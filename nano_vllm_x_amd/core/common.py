"""Request-level types shared by the scheduler, the engine and the model runner.

Field names and meanings follow nanovllmx/core/common.py (SamplingParams :4-16, outputs :22-37,
Sequence :44-54, ForwardBatch :61-65) so that callers written against the reference keep working.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Optional


class FinishReason(enum.Enum):
    STOP = enum.auto()      # the end-of-sequence token was produced
    LENGTH = enum.auto()    # a length cap was reached


class SequenceStatus(enum.Enum):
    WAITING = enum.auto()   # queued; owns no KV slots (or lost them to preemption)
    RUNNING = enum.auto()   # prefilled; owns one KV slot per token it has attended to
    FINISHED = enum.auto()


class ForwardMode(enum.Enum):
    PREFILL = enum.auto()
    DECODE = enum.auto()


@dataclass
class SamplingParams:
    n: int = 1
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1             # -1 = whole vocabulary
    min_p: float = 0.0
    max_tokens: Optional[int] = None       # cap on prompt + generated tokens (engine.py:149)
    max_new_tokens: Optional[int] = None   # cap on generated tokens (engine.py:151)
    stop: list = field(default_factory=list)
    ignore_eos: bool = False
    eos_token_id: int = -1


@dataclass(eq=False)
class Sequence:
    """One request.  Identity semantics: two requests are the same only if they are the same object
    (queues are searched by object; the reference's field-wise comparison gives the same answers
    because sequence ids are unique, at O(tokens) per comparison)."""
    seq_id: str
    status: SequenceStatus = SequenceStatus.WAITING
    num_tokens: int = 0
    prompt_len: int = 0
    token_ids: list = field(default_factory=list)
    sampling_params: SamplingParams = field(default_factory=SamplingParams)
    kv_indices: list = field(default_factory=list)   # slot of token i, for every token that has a KV row
    cached_kv_len: int = 0                           # tokens whose KV rows are already written

    # plain class attributes (not dataclass fields), set per object where they apply
    partial_of = None      # chunked prefill: the sequence this truncated copy stands for
    length_limit = None    # engine cache: total length at which a LENGTH stop rule fires (0 = none)

    @property
    def num_new_tokens(self) -> int:
        """Tokens the next forward pass must process."""
        return len(self.token_ids) - self.cached_kv_len


@dataclass
class ForwardBatch:
    forward_mode: ForwardMode
    num_seqs: int
    seqs: list


@dataclass
class EngineOutput:
    seq_id: str
    new_token_id: int
    is_finished: bool
    finish_reason: Optional[FinishReason]
    num_prompt_tokens: int
    num_generated_tokens: int


@dataclass
class GenerateOutput:
    token_str: str
    is_finished: bool
    finish_reason: Optional[FinishReason]
    num_prompt_tokens: int
    num_generated_tokens: int
    token_id: int = -1   # extension: the raw id, for callers that run without a tokenizer
