"""The slot-reuse argument of the direct-peer all-reduce (csrc/peer_allreduce.hip), checked on a model of it.

The kernel keeps TWO landing slots per (source rank, row), picked by the parity of the row's exchange count, and
never waits before it overwrites one.  That is safe only if a rank cannot reach exchange e + 2 of a row while a
peer still has to read exchange e of it.  Here every rank is a little state machine that does exactly what the
kernel does -- per exchange and row: push the tagged row into every peer's slot, poll its own slots until every
peer's tag is the current one, read them; the next exchange (the next launch on the stream) starts when all rows
of this one are done -- and a random scheduler interleaves the ranks' steps in every order it can, including a
rank running a whole exchange ahead.  Every read must return the payload of the exchange it belongs to.
No GPU: this is the host-side check of the protocol, the device tests are tests/test_gpu_peer.py.
"""
from __future__ import annotations

import random

import pytest


class Rank:
    def __init__(self, r, tp, max_rows, plan):
        self.r, self.tp, self.plan = r, tp, plan
        self.epoch = [0] * max_rows                       # per-row exchange count (device memory in the kernel)
        # landing buffer: slot[parity][source][row] = (tag, payload)
        self.slot = [[[(0, None)] * max_rows for _ in range(tp)] for _ in range(2)]
        self.call = 0                                     # index of the exchange (launch) in progress
        self.rows_todo: dict = {}                         # row -> "push" | "poll" for the launch in progress
        self.reads = []

    def start_launch(self):
        rows = self.plan[self.call]
        self.rows_todo = {i: "push" for i in range(rows)}

    def runnable(self, ranks):
        """(row, action) pairs that can make progress now."""
        out = []
        for i, st in self.rows_todo.items():
            e = self.epoch[i] + 1
            if st == "push":
                out.append((i, "push"))
            elif all(self.slot[e & 1][p][i][0] == e for p in range(self.tp) if p != self.r):
                out.append((i, "read"))
        return out

    def step(self, i, action, ranks):
        e = self.epoch[i] + 1
        if action == "push":
            for p in range(self.tp):
                if p != self.r:
                    ranks[p].slot[e & 1][self.r][i] = (e, (self.r, self.call, i))     # overwrites without asking
            self.rows_todo[i] = "poll"
        else:
            for p in range(self.tp):
                if p != self.r:
                    tag, payload = self.slot[e & 1][p][i]
                    assert tag == e and payload == (p, self.call, i), \
                        f"rank {self.r} exchange {self.call} row {i}: read {payload} with tag {tag} from rank {p}"
            self.reads.append((self.call, i))
            self.epoch[i] = e
            del self.rows_todo[i]
            if not self.rows_todo:                        # launch complete: the next one may start
                self.call += 1
                if self.call < len(self.plan):
                    self.start_launch()


@pytest.mark.parametrize("tp", [2, 3, 4, 8])
def test_two_slots_per_row_are_enough_under_every_interleaving_tried(tp):
    rng = random.Random(tp)
    for trial in range(60):
        max_rows = rng.randint(1, 5)
        plan = [rng.randint(1, max_rows) for _ in range(rng.randint(3, 25))]     # rows of consecutive exchanges, same on every rank
        ranks = [Rank(r, tp, max_rows, plan) for r in range(tp)]
        for k in ranks:
            k.start_launch()
        favourite = rng.randrange(tp)        # a rank the scheduler prefers: it runs ahead as far as the protocol lets it
        guard = 0
        while any(k.call < len(plan) for k in ranks):
            guard += 1
            assert guard < 200000, "the model made no progress: deadlock"
            choices = [(k, row, act) for k in ranks for row, act in k.runnable(ranks)]
            assert choices, "nobody can move: deadlock"
            fav = [c for c in choices if c[0].r == favourite]
            k, row, act = rng.choice(fav) if fav and rng.random() < 0.8 else rng.choice(choices)
            k.step(row, act, ranks)
            # the distance argument itself: no rank is ever two exchanges ahead of a peer on a row both have used
            lead = max(x.call for x in ranks) - min(x.call for x in ranks)
            assert lead <= 1, f"a rank got {lead} launches ahead of a peer"
        for k in ranks:
            assert len(k.reads) == sum(plan)


def test_one_slot_per_row_would_not_be_enough():
    """The same model with a single slot per row loses data: the check above is able to fail."""

    class OneSlot(Rank):
        def step(self, i, action, ranks):
            e = self.epoch[i] + 1
            if action == "push":
                for p in range(self.tp):
                    if p != self.r:
                        ranks[p].slot[0][self.r][i] = (e, (self.r, self.call, i))
                self.rows_todo[i] = "poll"
                return
            self.epoch[i] = e
            del self.rows_todo[i]
            self.call += 1
            self.start_launch()

        def runnable(self, ranks):
            out = []
            for i, st in self.rows_todo.items():
                e = self.epoch[i] + 1
                if st == "push":
                    out.append((i, "push"))
                elif all(self.slot[0][p][i][0] >= e for p in range(self.tp) if p != self.r):
                    out.append((i, "read"))
            return out

    plan = [1] * 6
    ranks = [OneSlot(r, 2, 1, plan) for r in range(2)]
    for k in ranks:
        k.start_launch()
    # rank 0 pushes exchange 0; rank 1 pushes, reads, and -- one launch ahead -- pushes exchange 1 over the slot
    # rank 0 has not read yet
    ranks[0].step(0, "push", ranks)
    ranks[1].step(0, "push", ranks)
    ranks[1].step(0, "read", ranks)
    ranks[1].step(0, "push", ranks)
    tag, payload = ranks[0].slot[0][1][0]
    assert payload == (1, 1, 0) and tag == 2, "rank 0's unread row of exchange 0 has been overwritten"
