"""The packed event merge (csrc/pymarshal.c: tm_new / tm_add / tm_finish) against the Python classes that restate the reference's
loop (transkun_amd.transcribe.EventMerger + resolveOverlapping; ModelTransformer.py:803-843, Data.py:170-214), on random event
streams: several recordings in lock step, recordings dropping out, events cut by the segment boundary (no onset / no offset),
overlapping events, both velocity types.  No GPU."""
import numpy as np
import pytest

from transkun_amd.transcribe import EventMerger, Note, PackedEventMerger

PITCHES = [-64, -67] + list(range(21, 33))


def _random_steps(rng, n_files, n_steps, P):
    """Per step: active files and, per active file and symbol, a time-ordered run of events (the order the device writes them)."""
    steps = []
    for s in range(n_steps):
        active = [f for f in range(n_files) if s < n_steps - (f % 3)]          # some recordings end earlier
        rows = []
        for a, f in enumerate(active):
            for sym in range(P):
                t = 8.0 * s - 8.0 + rng.uniform(0, 3)
                for _ in range(rng.integers(0, 5)):
                    start = t + rng.uniform(0, 3.0)
                    end = start + rng.uniform(1e-3, 6.0)
                    rows.append([start, end, float(rng.random() < 0.8), float(rng.random() < 0.8), float(rng.integers(0, 128)), sym,
                                 a * P + sym])
                    t = end
        steps.append((active, np.asarray(rows, dtype=np.float64).reshape(-1, 7)))
    return steps


@pytest.mark.parametrize("merge,resolve,vel_float,eager", [(True, True, False, False), (True, False, False, False), (False, True, True, False),
                                                           (True, True, True, False), (True, True, False, True), (True, False, True, True),
                                                           (True, True, True, True), (False, True, False, True)])
def test_packed_merge_equals_python_merge(merge, resolve, vel_float, eager):
    """eager: the Notes are made inside add_step as events settle (without the merge the request is ignored: same results)."""
    rng = np.random.default_rng(7)
    P = len(PITCHES)
    n_files, n_steps = 4, 6
    steps = _random_steps(rng, n_files, n_steps, P)
    packed = PackedEventMerger(n_files, PITCHES, merge, eager_resolve=resolve if eager else None)
    assert packed.eager == (eager and merge)
    packed.vel_float = vel_float
    ref = [EventMerger(merge) for _ in range(n_files)]
    for s, (active, rows) in enumerate(steps):
        nxt = steps[s + 1][0] if s + 1 < len(steps) else []
        packed.add_step(s, rows, rows.shape[0], active, later_events_from=[8.0 * s if f in nxt else 1e300 for f in active] if eager else None)
        # the reference's order within a segment: sorted by (start, end, pitch) (transcribeFrames :722)
        per_file = [[] for _ in active]
        for r in rows:
            sym, chain = int(r[5]), int(r[6])
            vel = float(r[4]) if vel_float else int(r[4])
            per_file[chain // P].append(Note(float(r[0]), float(r[1]), PITCHES[sym], vel, bool(r[2]), bool(r[3])))
        for a, f in enumerate(active):
            per_file[a].sort(key=lambda n: (n.start, n.end, n.pitch))
            ref[f].add_segment(per_file[a])
    for f in range(n_files):
        got = packed.finish(f, resolve)
        want = ref[f].finish(resolve)
        assert len(got) == len(want) and len(want) > 0
        assert [n.astuple() for n in got] == [n.astuple() for n in want]
        assert all(type(a.velocity) is type(b.velocity) and type(a.pitch) is int for a, b in zip(got, want))


def test_packed_merge_argument_errors():
    m = PackedEventMerger(2, PITCHES, True)
    rows = np.zeros((1, 7)); rows[0, 5] = 99                          # symbol out of range
    with pytest.raises(IndexError):
        m.add_step(0, rows, 1, [0])
    with pytest.raises(IndexError):
        m.add_step(0, np.zeros((1, 7)), 1, [5])                       # recording out of range
    assert m.finish(0) == [] and m.finish(1) == []
