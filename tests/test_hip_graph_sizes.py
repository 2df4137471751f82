"""The holder of captured decode graphs: ladder of batch sizes, matching a batch to a size, forgetting a size whose
graph turned out slower than a larger one's (ModelRunner.prune_slow_graph_sizes).  No GPU: the graphs are stand-ins."""
import pytest

from nano_vllm_x_amd.core.hip_graph import HIPGraph, decode_batch_ladder


def _holder(sizes):
    g = HIPGraph()
    for bs in sizes:
        g.graphs[bs] = object()
        g.captured_bs.append(bs)
        g.max_bs = max(g.max_bs, bs)
    return g


def test_ladder_matches_the_reference_rule():
    # 1, 2, 4, 8, then every 16 up to max_bs, max_bs itself always (nanovllmx/core/model_runner.py:335-337)
    assert decode_batch_ladder(50) == [1, 2, 4, 8, 16, 32, 48, 50]
    assert decode_batch_ladder(256)[-3:] == [224, 240, 256] and len(decode_batch_ladder(256)) == 20
    assert decode_batch_ladder(3) == [1, 2, 3]


def test_a_dropped_size_sends_its_batches_to_the_next_larger_graph():
    g = _holder([1, 2, 4, 8, 16, 32, 48])
    assert g.match_bs(9) == 16 and g.match_bs(16) == 16 and g.match_bs(17) == 32
    g.drop(16)
    assert g.captured_bs == [1, 2, 4, 8, 32, 48]
    assert g.match_bs(9) == 32 and g.match_bs(16) == 32 and g.match_bs(8) == 8
    g.drop(32)
    assert g.match_bs(9) == 48 and g.max_bs == 48
    with pytest.raises(AssertionError):
        g.drop(48)          # the largest size is what every batch can fall back on
    with pytest.raises(AssertionError):
        g.drop(16)          # already gone
