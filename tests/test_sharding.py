"""Multi-GPU path on CPU: contiguous window shards, two gloo ranks, ordered gather == single-process result.
(The per-rank worker here is the oracle, standing in for the GPU engine that this box does not have.)"""
import os
import subprocess
import sys

import pytest

from consent_amd.sharding import shard_by_cost, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_cost_is_contiguous_and_balanced():
    costs = [150, 3, 3, 150, 30, 30, 30, 30, 150, 5, 5, 90]
    for world in (1, 2, 4):
        spans = shard_by_cost(costs, world)
        assert spans[0][0] == 0 and spans[-1][1] == len(costs) and len(spans) == world
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    two = shard_by_cost(costs, 2)
    loads = [sum(costs[lo:hi]) for lo, hi in two]
    assert abs(loads[0] - loads[1]) <= max(costs)


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
import consent_amd as ca
from consent_amd.engine import synth_host
from consent_amd.sharding import shard_range, gather_in_order
import oracle_lib
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = 10
lo, hi = shard_range(N, rank, world)
prm = ca.Params(9, 4, 8, 2, 20)
spec = ca.SynthSpec.pacbio(hi - lo, 8, first_window=lo)      # each rank synthesises only its own windows
res, _ = oracle_lib.oracle_run(prm, synth_host(spec))
mine = [(lo + i, res.consensus(i), int(res.status[i])) for i in range(hi - lo)]
dist.barrier()
allr = gather_in_order(mine, rank, world, dist)
if rank == 0:
    full, _ = oracle_lib.oracle_run(prm, synth_host(ca.SynthSpec.pacbio(N, 8)))
    assert [w for w, _, _ in allr] == list(range(N))
    for w, cons, st in allr:
        assert cons == full.consensus(w) and st == int(full.status[w]), w
    print("SHARD_OK")
dist.destroy_process_group()
"""


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29617", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "SHARD_OK" in out.stdout
