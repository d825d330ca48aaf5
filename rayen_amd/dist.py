"""Batch-sharded projection across ranks (one process per GPU, ``torch.distributed``).

The reference has no parallelism of any kind.  Samples are
independent, so the batch dimension shards with NO data-path collective: every
rank projects its own rows with the same replicated constants (<= a few hundred
KiB).  A caller that needs every output on every rank (the layout BASELINE.json's
north_star describes) adds exactly one collective, an all-gather of ``y`` (RCCL
over xGMI when the backend is ``nccl``); it can be issued in chunks so that the
gather of chunk c overlaps the projection of chunk c+1.

``project_fn`` is the per-rank compute (by default the ``ConstraintModule``
itself).  It is a parameter so that the sharding/gather logic can be exercised
on CPU with the ``gloo`` backend, where the HIP kernels cannot run.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous row block of ``rank``: the first ``total % world`` ranks get one extra row."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int):
    return [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]


class ShardedProjection:
    """Data-parallel wrapper around a projection callable ``[b, ...] -> [b, k, 1]``."""

    def __init__(self, project_fn, group=None):
        self.project_fn = project_fn
        self.group = group

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def forward_local(self, x_local):
        """This rank's rows only; no communication (the data-parallel training case)."""
        return self.project_fn(x_local)

    def forward_replicated(self, x_full, chunks: int = 1):
        """``x_full`` is replicated on every rank: project this rank's slice, all-gather ``y``.

        Returns the full ``[B, k, 1]`` result on every rank, rows in the original order.
        """
        lo, hi = shard_bounds(x_full.shape[0], self.world, self.rank)
        return self.all_gather_rows(self.project_fn, x_full[lo:hi], x_full.shape[0], chunks)

    def forward_gather(self, x_local, chunks: int = 1):
        """Every rank holds its own rows (equal counts or not): project them, all-gather ``y``."""
        if self.world == 1:
            return self.project_fn(x_local)
        counts = torch.zeros(self.world, dtype=torch.int64, device=x_local.device)
        counts[self.rank] = x_local.shape[0]
        dist.all_reduce(counts, group=self.group)
        return self.all_gather_rows(self.project_fn, x_local, int(counts.sum().item()), chunks,
                                    sizes=[int(c) for c in counts.tolist()])

    # ------------------------------------------------------------------
    def all_gather_rows(self, fn, x_local, total, chunks=1, sizes=None):
        world = self.world
        if world == 1:
            return fn(x_local)
        sizes = sizes or shard_sizes(total, world)
        max_rows = max(sizes)
        n_local = x_local.shape[0]
        chunks = max(1, min(chunks, max_rows))
        step = -(-max_rows // chunks)
        out = None
        handles = []
        pieces = []  # (chunk index, per-rank receive buffers)
        for c in range(chunks):
            lo, hi = min(c * step, n_local), min((c + 1) * step, n_local)
            rows = min((c + 1) * step, max_rows) - c * step
            if rows <= 0:
                break
            y_c = fn(x_local[lo:hi])  # may be an empty slice on a rank with a shorter shard
            if out is None:
                tail = tuple(y_c.shape[1:])
                out = torch.empty((total,) + tail, dtype=y_c.dtype, device=y_c.device)
            send = torch.zeros((rows,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
            send[: y_c.shape[0]] = y_c
            recv = [torch.empty_like(send) for _ in range(world)]
            handles.append(dist.all_gather(recv, send, group=self.group, async_op=True))
            pieces.append((c, recv))
        offsets = [0]
        for s in sizes:
            offsets.append(offsets[-1] + s)
        for handle, (c, recv) in zip(handles, pieces):
            handle.wait()
            for r in range(world):
                lo = min(c * step, sizes[r])
                hi = min((c + 1) * step, sizes[r])
                if hi > lo:
                    out[offsets[r] + lo: offsets[r] + hi] = recv[r][: hi - lo]
        return out
