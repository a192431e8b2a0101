"""World-size-2 gloo test of the batch sharding + result gather (the N > 1 path of bench.py /
smplfitter_amd.dist).  The per-rank fit is stood in by the CPU oracle on a tiny batch so that the
test runs without a GPU; the distributed plumbing under test is identical for nccl/RCCL."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from smplfitter_amd import dist as sd

    J, S = 24, 10
    rs = np.random.RandomState(0)
    full = torch.from_numpy(rs.randn(total, 3 * J + S + 3).astype(np.float32))

    def fake_fit(tv, tj, **kw):  # rows of `full` keyed by the first coordinate of the "vertices"
        idx = tv[:, 0, 0].long()
        rows = full[idx]
        return sd.unpack_results(rows, J, S)

    tv = torch.arange(total, dtype=torch.float32).view(total, 1, 1).expand(total, 2, 3).contiguous()
    out = sd.fit_sharded(fake_fit, tv, None, J, S)
    got = sd.pack_results(out)
    ok = torch.equal(got, full)
    lo, hi = sd.shard_range(total, rank, world)
    torch.save(dict(ok=ok, lo=lo, hi=hi), os.path.join(tmp, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [8, 7])
def test_shard_and_gather_world2(total, tmp_path):
    port = 29500 + (os.getpid() % 1000) + total
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'r0.pt')
    r1 = torch.load(tmp_path / 'r1.pt')
    assert r0['ok'] and r1['ok']
    assert r0['lo'] == 0 and r0['hi'] == r1['lo'] and r1['hi'] == total


def test_shard_range_covers():
    from smplfitter_amd.dist import shard_range

    for total in (0, 1, 7, 262144):
        for world in (1, 2, 4, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
