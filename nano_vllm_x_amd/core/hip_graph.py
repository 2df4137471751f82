"""Holder of captured decode graphs, one per padded batch size (role of nanovllmx/core/cuda_graph.py).

On ROCm `torch.cuda.CUDAGraph` is a hipGraph: capture goes through hipStreamBeginCapture on the
current stream, and every kernel of libnvx_hip.so is launched on that stream with no allocation or
synchronisation of its own, so the whole decode step -- block-table update, split plan, the model's
layers, the RCCL all-reduces -- becomes one hipGraphLaunch.
"""
from __future__ import annotations

import bisect
from contextlib import contextmanager

import torch


class HIPGraph:
    def __init__(self):
        self.graphs: dict = {}
        self.captured_bs: list = []
        self.input_buffers: dict = {}
        self.output_buffers: dict = {}
        self.graph_pool = None
        self.max_bs = 0

    @property
    def is_captured(self) -> bool:
        return bool(self.graphs)

    def set_input_buffer(self, name: str, tensor: torch.Tensor):
        self.input_buffers[name] = tensor

    def get_input_buffer(self, name: str) -> torch.Tensor:
        return self.input_buffers[name]

    def set_output_buffer(self, name: str, tensor: torch.Tensor):
        self.output_buffers[name] = tensor

    def get_output_buffer(self, name: str) -> torch.Tensor:
        return self.output_buffers[name]

    @contextmanager
    def capture(self, bs: int):
        """Record everything launched inside the block as the graph for batch size `bs`; graphs
        share one memory pool, so capture the largest first (cuda_graph.py:27-39)."""
        g = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread makes runtime calls of its own; "thread_local" keeps
        # those from invalidating the capture going on in this thread
        mode = "thread_local" if torch.distributed.is_available() and torch.distributed.is_initialized() else "global"
        with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode=mode):
            yield
        if self.graph_pool is None:
            self.graph_pool = g.pool()
        self.graphs[bs] = g
        bisect.insort(self.captured_bs, bs)
        self.max_bs = max(self.max_bs, bs)

    def replay(self, bs: int):
        self.graphs[bs].replay()

    def match_bs(self, bs: int) -> int:
        """Smallest captured batch size that holds `bs` rows (cuda_graph.py:44-47)."""
        assert self.graphs and bs <= self.max_bs
        return self.captured_bs[bisect.bisect_left(self.captured_bs, bs)]

    def drop(self, bs: int):
        """Forget the graph of one batch size: batches of that size then replay the next larger one."""
        assert bs in self.graphs and bs != self.max_bs
        del self.graphs[bs]
        self.captured_bs.remove(bs)

    def clear(self):
        self.graphs.clear()
        self.captured_bs.clear()
        self.input_buffers.clear()
        self.output_buffers.clear()
        self.graph_pool = None
        self.max_bs = 0


def decode_batch_ladder(max_bs: int) -> list:
    """Batch sizes to capture: 1, 2, 4, 8, then every 16 up to max_bs (model_runner.py:335-337)."""
    sizes = [b for b in (1, 2, 4, 8) if b <= max_bs] + list(range(16, max_bs, 16))
    if max_bs not in sizes:
        sizes.append(max_bs)
    return sizes
