"""Batch sharding of the fit across the GPUs of one node (one process per GPU).

The fit of every instance is independent (reference: all reductions in ``BodyFitter.fit`` are
intra-instance unless ``share_beta``), so rank ``r`` of ``W`` fits a contiguous block of rows and the
only communication is ONE all-gather of the packed result rows ``(B_local, 3J+S+3 [+ optional columns])`` —
``torch.distributed`` backend ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.

``share_beta`` is the one mode with an exchange step inside the fit (reference pt/lstsq.py:24-26: the
normal equations are summed over the batch): every shape solve all-reduces the ``S*S + S`` doubles of
the rank-local sums (``smplfit_fit_args.share_allreduce``), ``num_iter`` tiny collectives per fit.
"""

from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of ``total`` rows for ``rank``; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# result columns in packing order; the optional ones are packed when the fit returned them
_OPTIONAL = ('kid_factor', 'scale_corr', 'orientations', 'relative_orientations')


def pack_results(res: dict) -> tuple[torch.Tensor, list]:
    """Rows ``pose_rotvecs | shape_betas | trans`` followed by every optional per-instance result the fit
    returned (``kid_factor``, ``scale_corr``, ``orientations``, ``relative_orientations``, flattened), and
    the layout ``[(key, trailing shape), ...]`` needed to unpack them."""
    keys = ['pose_rotvecs', 'shape_betas', 'trans'] + [k for k in _OPTIONAL if res.get(k) is not None]
    layout = [(k, tuple(res[k].shape[1:])) for k in keys]
    B = res['pose_rotvecs'].shape[0]
    return torch.cat([res[k].reshape(B, -1) for k in keys], dim=1).contiguous(), layout


def unpack_results(rows: torch.Tensor, layout: list) -> dict:
    out, c = {}, 0
    for key, shape in layout:
        n = 1
        for d in shape:
            n *= d
        out[key] = rows[:, c:c + n].reshape(rows.shape[0], *shape)
        c += n
    return out


def gather_rows(local_rows: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather ragged row blocks laid out by ``shard_range`` into one ``(total, C)`` tensor."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if local_rows.is_cuda and dist.get_backend(group) == 'gloo':  # host collective: stage explicitly
        return gather_rows(local_rows.cpu(), total, group).to(local_rows.device)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    assert local_rows.shape[0] == sizes[rank]
    cols = local_rows.shape[1]
    if len(set(sizes)) == 1:  # even shards: one fused collective
        out = torch.empty((total, cols), dtype=local_rows.dtype, device=local_rows.device)
        dist.all_gather_into_tensor(out, local_rows.contiguous(), group=group)
        return out
    mx = max(sizes)
    padded = torch.zeros((mx, cols), dtype=local_rows.dtype, device=local_rows.device)
    padded[: sizes[rank]] = local_rows
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


class OverlappedGather:
    """The result gather of step ``k`` behind the fit of step ``k + 1``.

    A serving / benchmark loop that fits one batch per step needs the gathered rows of a step only after the next step's
    fit has been enqueued, so the collective does not have to sit on the fit's stream (at 4096 fits per GPU a step is
    1.5 ms: an 8-rank all-gather issued in line comes straight off the scaling efficiency).  ``submit(parts)`` records an
    event on the caller's stream — nothing else: no kernel, no wait — and a SIDE stream, behind that event, packs the
    step's result tensors into one of ``depth`` row buffers and runs ``all_gather_into_tensor`` (RCCL orders the
    collective after the side stream's work); the caller's stream goes on with the next fit at once.  A buffer is reused
    ``depth`` steps later, in order on the side stream; the gathered rows of a slot must be consumed before its reuse.  ``result()`` makes the caller's stream wait for the latest collective and
    returns its ``(world * B, C)`` rows.  CPU tensors (the gloo tests): ``async_op=True`` work handles play the part of
    the events.  Equal row counts on every rank (the benchmark's weak-scaling layout)."""

    def __init__(self, rows: int, cols: int, dtype=torch.float32, device='cpu', group=None, depth: int = 2):
        self.group, self.depth, self.k = group, depth, 0
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.host_staged = self.cuda and dist.get_backend(group) == 'gloo'  # tests on one GPU: no device collective
        self.send = [torch.empty((rows, cols), dtype=dtype, device=self.device) for _ in range(depth)]
        self.recv = [torch.empty((self.world * rows, cols), dtype=dtype, device=self.device) for _ in range(depth)]
        self.work = [None] * depth
        if self.cuda:
            self.side = torch.cuda.Stream(device=self.device)
            self.ready = [torch.cuda.Event() for _ in range(depth)]   # rows of the slot are packed (caller's stream)
            self.done = [torch.cuda.Event(enable_timing=True) for _ in range(depth)]   # its collective has finished (side stream)
            self.start = [torch.cuda.Event(enable_timing=True) for _ in range(depth)]
            self.used = [False] * depth

    def submit(self, parts) -> int:
        """``parts``: the step's result tensors ``(B, c_i)`` in packing order (or one packed tensor); returns the slot."""
        s = self.k % self.depth
        self.k += 1
        parts = [parts] if isinstance(parts, torch.Tensor) else list(parts)
        if self.cuda:
            # the caller's stream only records an event: packing, the wait for the slot's previous collective (in order on
            # the side stream) and the collective all run on the side stream
            cur = torch.cuda.current_stream(self.device)
            self.ready[s].record(cur)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready[s])
                for p in parts:
                    p.record_stream(self.side)  # (the caching allocator must not hand the fit's outputs out again before the pack has read them)
                self.start[s].record(self.side)
                torch.cat(parts, dim=1, out=self.send[s])
                if self.host_staged:
                    hg = torch.empty(self.recv[s].shape, dtype=self.recv[s].dtype)
                    dist.all_gather_into_tensor(hg, self.send[s].cpu(), group=self.group)
                    self.recv[s].copy_(hg)
                else:
                    dist.all_gather_into_tensor(self.recv[s], self.send[s], group=self.group)
                self.done[s].record(self.side)
            self.used[s] = True
        else:
            if self.work[s] is not None:
                self.work[s].wait()
            torch.cat(parts, dim=1, out=self.send[s])
            self.work[s] = dist.all_gather_into_tensor(self.recv[s], self.send[s], group=self.group, async_op=True)
        return s

    def result(self, slot: Optional[int] = None) -> torch.Tensor:
        """The gathered rows of ``slot`` (default: the latest submitted step), valid on the caller's stream."""
        s = (self.k - 1) % self.depth if slot is None else slot
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.done[s])
        elif self.work[s] is not None:
            self.work[s].wait()
            self.work[s] = None
        return self.recv[s]

    def gather_ms(self, slot: int) -> float:
        """HIP-event time of the slot's collective on the side stream (after a device synchronisation)."""
        return self.start[slot].elapsed_time(self.done[slot]) if self.cuda else 0.0

    def finish(self) -> None:
        for s in range(self.depth):
            if self.cuda:
                if self.used[s]:
                    self.done[s].synchronize()
            elif self.work[s] is not None:
                self.work[s].wait()
                self.work[s] = None


def fit_sharded(fit_fn, target_vertices: torch.Tensor, target_joints: Optional[torch.Tensor],
                num_joints: int, num_betas: int, group=None, **fit_kwargs) -> dict:
    """Every rank holds the FULL ``(B, V, 3)`` inputs (or generates them); each fits its block with
    ``fit_fn`` (e.g. ``BodyFitter.fit``) and all ranks receive the full result dict."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    total = target_vertices.shape[0]
    if fit_kwargs.get('share_beta'):
        # the shared shape couples every instance of the batch in each solve: one all-reduce of the
        # (S*S + S) summed systems per shape solve, issued from inside the fit
        if total < world:
            raise ValueError('a sharded share_beta fit needs at least one instance per rank')
        fit_kwargs = dict(fit_kwargs, share_beta_group=group if group is not None else dist.group.WORLD)
    lo, hi = shard_range(total, rank, world)
    res = fit_fn(target_vertices[lo:hi], None if target_joints is None else target_joints[lo:hi],
                 **fit_kwargs)
    rows, layout = pack_results(res)  # the layout is the same on every rank (same options)
    assert layout[0] == ('pose_rotvecs', (3 * num_joints,)) and layout[1][1] == (num_betas,), layout
    return unpack_results(gather_rows(rows, total, group), layout)
