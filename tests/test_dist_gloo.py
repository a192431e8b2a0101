"""World-size-2 gloo tests of the batch sharding + result gather (the N > 1 path of bench.py /
smplfitter_amd.dist) and of the one exchange step the path has: the all-reduce of the summed normal
equations in a sharded ``share_beta`` fit.  The per-rank fit is stood in by a table lookup resp. the
host-compiled stage code so that the tests run without a GPU; the distributed plumbing under test
(``fit_sharded``, the ``share_allreduce`` callback contract) is identical for nccl/RCCL."""

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
    # pose | betas | trans | kid_factor | scale_corr | orientations: the optional columns travel as well
    layout = [('pose_rotvecs', (3 * J,)), ('shape_betas', (S,)), ('trans', (3,)), ('kid_factor', ()),
              ('scale_corr', ()), ('orientations', (J, 3, 3))]
    width = 3 * J + S + 3 + 1 + 1 + 9 * J
    full = torch.from_numpy(rs.randn(total, width).astype(np.float32))

    def fake_fit(tv, tj, **kw):  # rows of `full` keyed by the first coordinate of the "vertices"
        idx = tv[:, 0, 0].long()
        return sd.unpack_results(full[idx], layout)

    tv = torch.arange(total, dtype=torch.float32).view(total, 1, 1).expand(total, 2, 3).contiguous()
    out = sd.fit_sharded(fake_fit, tv, None, J, S)
    got, got_layout = sd.pack_results(out)
    ok = torch.equal(got, full) and got_layout == layout and out['orientations'].shape == (total, J, 3, 3) \
        and out['kid_factor'].shape == (total,)
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


def _overlap_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from smplfitter_amd import dist as sd

    B, J, S, steps = 5, 24, 10, 7
    og = sd.OverlappedGather(B, 3 * J + S + 3, device='cpu', depth=2)

    def rows_of(step, r):  # what rank r's fit of `step` returns: three result tensors
        g = torch.Generator().manual_seed(1000 * step + r)
        return [torch.randn(B, 3 * J, generator=g), torch.randn(B, S, generator=g), torch.randn(B, 3, generator=g)]

    ok = True
    prev = None
    for k in range(steps):
        slot = og.submit(rows_of(k, rank))   # step k's gather is in flight ...
        if prev is not None:                 # ... while step k - 1's rows are consumed
            pk, pslot = prev
            got = og.result(pslot)
            want = torch.cat([torch.cat(rows_of(pk, r), dim=1) for r in range(world)], dim=0)
            ok = ok and torch.equal(got, want)
        prev = (k, slot)
    got = og.result()
    want = torch.cat([torch.cat(rows_of(steps - 1, r), dim=1) for r in range(world)], dim=0)
    ok = ok and torch.equal(got, want)
    og.finish()
    torch.save(dict(ok=bool(ok)), os.path.join(tmp, f'o{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_world2(tmp_path):
    """The double-buffered result gather of bench.py --gpus N (dist.OverlappedGather): step k's all-gather in flight while
    step k + 1 packs its rows; every step's gathered rows are the ranks' rows of THAT step, in rank order."""
    port = 29500 + (os.getpid() % 1000) + 77
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert torch.load(tmp_path / 'o0.pt')['ok'] and torch.load(tmp_path / 'o1.pt')['ok']


def _share_worker(rank, world, port, root, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import hostemu_util as H
    import util
    from smplfitter_amd import dist as sd

    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_smpl.npz'), allow_pickle=False))
    kind, md = util.load_md(root, 'smpl', g)
    om, _ = util.make_oracle(md, kind)
    _, tv, kw = util.share_inputs(g, om, 'a')
    tj = kw.pop('target_joints')
    calls = []

    def fit_fn(v, j, share_beta_group=None, **k):  # BodyFitter.fit's contract, host-compiled stages
        def allreduce(sums):
            t = torch.from_numpy(sums)  # shares memory with the C buffer
            dist.all_reduce(t, group=share_beta_group)
            calls.append(1)

        o = H.fit_warm(md, kind, v.numpy(), target_joints=None if j is None else j.numpy(),
                       share_allreduce=allreduce if share_beta_group is not None else None, **k)
        return {n: torch.from_numpy(a) for n, a in o.items()}

    out = sd.fit_sharded(fit_fn, torch.from_numpy(tv), torch.from_numpy(tj), md.num_joints, 10,
                         share_beta=True, **kw)
    if rank == 0:
        whole = H.fit_warm(md, kind, tv, target_joints=tj, share_beta=True, **kw)
        np.savez(os.path.join(tmp, 'share.npz'), ncalls=len(calls), num_iter=kw['num_iter'],
                 **{f'sharded_{n}': a.numpy() for n, a in out.items()},
                 **{f'whole_{n}': whole[n] for n in ('pose_rotvecs', 'shape_betas', 'trans')})
    dist.barrier()
    dist.destroy_process_group()


def test_share_beta_sharded_world2(model_root, tmp_path):
    """share_beta over a batch split between two ranks: each rank sums its own systems, the callback
    all-reduces them, both solve the same sum — the shape of the whole batch, as one rank computes it
    (the fp64 sums only differ in their order)."""
    port = 29500 + (os.getpid() % 1000) + 17
    mp.spawn(_share_worker, args=(2, port, model_root, str(tmp_path)), nprocs=2, join=True)
    r = np.load(tmp_path / 'share.npz')
    assert int(r['ncalls']) == int(r['num_iter'])  # one collective per shape solve
    betas = r['sharded_shape_betas']
    assert np.abs(betas - betas[:1]).max() == 0  # one shape on every rank's rows
    for n, tol in (('shape_betas', 2e-6), ('trans', 2e-6), ('pose_rotvecs', 2e-5)):
        assert np.abs(r[f'sharded_{n}'] - r[f'whole_{n}']).max() < tol, n


def test_shard_range_covers():
    from smplfitter_amd.dist import shard_range

    for total in (0, 1, 7, 262144):
        for world in (1, 2, 4, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
