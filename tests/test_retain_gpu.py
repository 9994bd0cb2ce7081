"""-m gpu: ssv_batch_retain / ssv_batch_release - device batches copied into memory that stays (what `seeksv run` and bench.py's file leg keep between the getclip
and the getsv passes), cut out of arenas: the kept copies give the tables of the batches they were made of, in any order of release, across several arenas."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORKER = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from seeksv_amd import synth
from seeksv_amd.device import Context
ctx = Context(0)
w = synth.Workload(genome_frac=1 / 2048, depth=30, n_sv=20)
cuts = np.linspace(0, w.n_total, 9).astype(np.int64)
made = [w.generate_device(int(cuts[i]), int(cuts[i + 1] - cuts[i]), 0) for i in range(8)]
want = ctx.getclip([b for b, _ in made])
kept = [ctx.batch_retain(b) for b, _ in made]          # ~3 MB each: with SSV_RETAIN_ARENA_MB=4 several arenas
del made                                              # (the kept copies stand on their own: hot columns, record lines, CIGARs, the soft-clipped reads' bases)
got = ctx.getclip(kept)
for name in ("tid", "pos", "side", "support", "left_len", "right_len", "str", "cigar"):
    assert np.array_equal(want[name], got[name]), name
assert want["n_clusters"] == got["n_clusters"] > 100
for i in (5, 0, 7, 2):                                   # some go, the others still give their tables
    ctx.batch_release(kept[i]); kept[i] = None
rest = [k for k in kept if k is not None]
part = ctx.getclip(rest)
assert 0 < part["n_clusters"] < want["n_clusters"]
again = [ctx.batch_retain(k) for k in rest]              # a copy of a kept batch, in room that was given back or in a new arena
assert ctx.getclip(again)["n_clusters"] == part["n_clusters"]
import ctypes
dup = type(again[0]).from_buffer_copy(again[0])         # a by-value copy of a handle: released once through the original ...
ctx.batch_release(again[0])
try:
    ctx.batch_release(dup)                                # ... and refused through the copy: its slab is no longer live (the arena's count stays right,
    raise SystemExit("the copy of a released handle was accepted")   # the batches that still live in it keep their memory)
except RuntimeError:
    pass
assert ctx.getclip(again[1:])["n_clusters"] > 0
for k in rest + again[1:]:
    ctx.batch_release(k)
bad = rest[0]
try:
    ctx.batch_release(bad)                                # twice: refused (the handle was cleared by the first release)
    raise SystemExit("a second release was accepted")
except RuntimeError:
    pass
ctx.close()
print("ok")
'''


@pytest.mark.parametrize("arena_mb", ["4", "4096"])
def test_retained_batches_in_arenas(arena_mb):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", WORKER % root], env=dict(os.environ, SSV_RETAIN_ARENA_MB=arena_mb), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
