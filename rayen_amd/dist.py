"""Batch-sharded projection across ranks (one process per GPU, ``torch.distributed``).

The reference has no parallelism of any kind.  Samples are
independent, so the batch dimension shards with NO data-path collective: every
rank projects its own rows with the same replicated constants (<= a few hundred
KiB).  A caller that needs every output on every rank (the layout BASELINE.json's
north_star describes) adds exactly one collective, an all-gather of ``y`` (RCCL
over xGMI when the backend is ``nccl``).

:class:`ShardedStep` is that step, and it is what ``bench.py --gpus N`` times:

* the local batch is cut into ``chunks`` row blocks; block ``c`` is projected
  STRAIGHT INTO its send buffer (``project_into(x_rows, out_rows)``: the HIP kernel
  writes the gather's input, no staging copy, no zero-fill),
* ``all_gather_into_tensor`` of block ``c`` is issued asynchronously as soon as its
  projection is queued -- RCCL runs it on its own stream, after the kernel, while
  the main stream already projects block ``c + 1`` (whether they really overlap depends on the
  projection's persistent grid leaving compute units for RCCL's kernels: ``reserve_cus``; ``trace=True``
  time-stamps every chunk's projection end and gather end so that a run shows it),
* the result is ONE buffer ``[chunks, world, rows, k]``; ``rows_of(rank)`` is a
  strided view of it in the rank's original row order (no reorder pass).

Arithmetic that sizes ``chunks`` (MI355X, direct xGMI mesh): a rank receives
``(world - 1) x B x k x 4`` bytes -- config 3 at 8 ranks: 7 x 64 MiB over 7 links --
while projecting ``B`` rows takes 0.1 ms; the gather is the longer leg by ~10x, so
chunking exists to start it early (after 1/chunks of the compute), not to hide it.

``project_into`` is a parameter so that exactly this code runs on CPU with the
``gloo`` backend in ``tests/test_dist_gloo.py`` (the HIP kernels cannot run there).

``gather_impl="peer"`` (round 6; ``bench.py --gather-impl peer``) replaces the collective by what an all-gather IS on a
direct xGMI mesh: every rank copies block ``c`` of its send buffer straight into slot ``[c, rank]`` of every peer's gather
buffer (``hipMemcpyPeerAsync``: SDMA engines, no compute units taken from the projection's persistent grid, one copy per
link) on a side stream behind the projection of that block, and one barrier at the end of the step tells every rank that
its own buffer is complete.  The peers' buffers are opened once, at construction, through :class:`CudaIpcBuffers` (the
``torch.multiprocessing`` CUDA-IPC reduction, dmabuf handles); ``tests/test_dist_gloo.py`` drives the same indexing on
the host with :class:`ShmBuffers` (every rank's buffer a file under ``/dev/shm`` that all ranks map).  Never run on more
than one GPU (no node was available): an A/B for the first multi-GPU run, RCCL stays the default.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class CudaIpcBuffers:
    """The gather buffers of all ranks, each opened in every process: ``allocate`` returns ``(own, views)`` with
    ``views[r]`` = rank ``r``'s buffer as a tensor of THIS process (``views[rank] is own``)."""

    def allocate(self, shape, dtype, device, rank, world, group):
        from torch.multiprocessing.reductions import reduce_tensor
        own = torch.empty(shape, dtype=dtype, device=device)
        rebuild, args = reduce_tensor(own)                     # (the CUDA-IPC handle of the allocation + the view's geometry)
        handles = [None] * world
        dist.all_gather_object(handles, (rebuild, args), group=group)
        views = [own if r == rank else handles[r][0](*handles[r][1]) for r in range(world)]
        self._keep = handles                                   # (the senders' reference counters live as long as the step)
        return own, views


class ShmBuffers:
    """Host stand-in for :class:`CudaIpcBuffers` (tests): rank ``r``'s buffer is the file ``<prefix>.<r>`` under ``/dev/shm``,
    mapped by every rank -- a write into ``views[r]`` lands in rank ``r``'s own tensor, as a peer copy does."""

    def __init__(self, prefix):
        self.prefix = prefix

    def allocate(self, shape, dtype, device, rank, world, group):
        import numpy as np
        np_dtype = {torch.float32: np.float32, torch.float64: np.float64}[dtype]
        count = 1
        for d in shape:
            count *= int(d)
        path = f"{self.prefix}.{rank}"
        np.memmap(path, dtype=np_dtype, mode="w+", shape=(max(count, 1),)).flush()
        dist.barrier(group=group)                              # every file exists
        self._maps = [np.memmap(f"{self.prefix}.{r}", dtype=np_dtype, mode="r+", shape=(max(count, 1),)) for r in range(world)]
        views = [torch.from_numpy(m)[:count].view(shape) for m in self._maps]
        return views[rank], views


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous row block of ``rank``: the first ``total % world`` ranks get one extra row."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int):
    return [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]


class ShardedStep:
    """One data-parallel step: project this rank's rows, optionally all-gather ``y``.

    ``sizes``: rows held by every rank (``shard_sizes(total, world)`` for a sharded batch, ``[B] * world``
    for fixed per-rank work).  Buffers are allocated once, here; a step allocates nothing.
    """

    def __init__(self, project_into, sizes, k, dtype, device, chunks=4, gather=True, group=None, gather_alone=False,
                 reserve_cus=0, set_reserve=None, gather_impl="rccl", peer_buffers=None):
        """``reserve_cus`` / ``set_reserve``: compute units the projection's persistent grid leaves free while a
        gather step runs (``set_reserve(n) -> previous`` = ``rayen_reserve_cus`` of the C ABI; ``None`` on CPU).
        RCCL's all-gather kernels need CUs: behind a grid that fills every SIMD they would simply queue.
        The reservation is PROCESS-WIDE state of the library for the duration of the step (``rayen_reserve_cus`` is one
        atomic): another thread or pack that launches a projection inside that window gets the smaller grid too (same
        results, fewer workgroups).  One stepping thread per process is the supported arrangement; the default is 0
        (no reservation) until a multi-GPU run shows that leaving CUs free pays."""
        if gather_impl not in ("rccl", "peer"):
            raise ValueError(f"gather_impl must be 'rccl' or 'peer', got {gather_impl!r}")
        self.project_into = project_into
        self.gather_impl = gather_impl
        self.peer_views = None
        self.reserve_cus = int(reserve_cus)
        self.set_reserve = set_reserve
        self.last_trace = None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if len(sizes) != self.world:
            raise ValueError(f"sizes has {len(sizes)} entries for {self.world} ranks")
        self.sizes = [int(s) for s in sizes]
        self.n_local = self.sizes[self.rank]
        self.k = int(k)
        # (gather_alone: issue the collective even in a one-rank group -- exercises the RCCL path on a single GPU)
        self.gather = bool(gather) and (self.world > 1 or (gather_alone and dist.is_initialized()))
        max_rows = max(self.sizes) if self.sizes else 0
        self.chunks = max(1, min(int(chunks), max_rows)) if self.gather else 1
        self.rows = -(-max_rows // self.chunks) if max_rows else 0          # rows per chunk (last one ragged)
        if self.gather:
            # [chunk][rank][row][k]: chunk c of every rank is one contiguous all-gather output
            shape = (self.chunks, self.world, self.rows, self.k)
            if gather_impl == "peer":
                # (gather_impl "peer": every rank's buffer opened in every process, once; the step only copies)
                buffers = peer_buffers if peer_buffers is not None else CudaIpcBuffers()
                self.out, self.peer_views = buffers.allocate(shape, dtype, device, self.rank, self.world, group)
                self._buffers = buffers
                self._copy_stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
            else:
                self.out = torch.empty(shape, dtype=dtype, device=device)
            self.send = torch.empty((self.chunks, self.rows, self.k), dtype=dtype, device=device)
        else:
            self.out = None
            self.send = torch.empty((1, max(self.n_local, 0), self.k), dtype=dtype, device=device)

    def __call__(self, x_local, trace=False):
        """``x_local [n_local, ...]`` -> this rank's ``y [n_local, k]`` (no gather) or the gather buffer.

        ``trace``: also time-stamp every chunk -- its projection queued / finished, its all-gather finished (as the
        main stream sees it) -- into ``self.last_trace`` (milliseconds from the start of the step; HIP events on a
        device, ``perf_counter`` on the CPU).  One synchronisation at the end; not for timed loops."""
        if x_local.shape[0] != self.n_local:
            raise ValueError(f"rank {self.rank} holds {self.n_local} rows, got {x_local.shape[0]}")
        if not self.gather:
            y = self.send[0]
            self.project_into(x_local, y)
            return y
        on_gpu = self.send.is_cuda
        stamps = []

        def stamp():
            if not trace:
                return None
            if on_gpu:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                return ev
            import time
            return time.perf_counter()

        prev = self.set_reserve(self.reserve_cus) if (self.set_reserve is not None and self.reserve_cus > 0) else None
        try:
            t0 = stamp()
            works = []
            for c in range(self.chunks):
                lo, hi = min(c * self.rows, self.n_local), min((c + 1) * self.rows, self.n_local)
                send = self.send[c]
                if hi > lo:
                    self.project_into(x_local[lo:hi], send[: hi - lo])      # rows beyond hi - lo: padding, never read
                projected = stamp()
                if self.gather_impl == "peer":
                    works.append(self._peer_copies(c, send))
                else:
                    works.append(dist.all_gather_into_tensor(self.out[c].view(self.world * self.rows, self.k), send,
                                                             group=self.group, async_op=True))
                stamps.append([projected, None])
            for c, work in enumerate(works):
                if work is not None:
                    work.wait()
                stamps[c][1] = stamp()
            if self.gather_impl == "peer":
                # every rank has issued (and waited for) ITS copies; the barrier says the others' have landed here too
                dist.barrier(group=self.group)
        finally:
            if prev is not None:
                self.set_reserve(prev)
        if trace:
            if on_gpu:
                torch.cuda.synchronize()
                ms = lambda ev: t0.elapsed_time(ev)                          # noqa: E731
            else:
                ms = lambda t: (t - t0) * 1e3                                # noqa: E731
            self.last_trace = [{"chunk": c, "rows": int(min((c + 1) * self.rows, self.n_local) - min(c * self.rows, self.n_local)),
                                "projection_end_ms": ms(a), "gather_end_ms": ms(b)} for c, (a, b) in enumerate(stamps)]
        return self.out

    def _peer_copies(self, c, send):
        """Block ``c`` of this rank into slot ``[c, rank]`` of every rank's buffer (its own included).  On a device: on the
        copy stream, behind the projection that produced the block; returns an object whose ``wait()`` makes the main stream
        wait for the copies."""
        if self._copy_stream is None:
            for view in self.peer_views:
                view[c, self.rank].copy_(send)
            return None
        main = torch.cuda.current_stream()
        self._copy_stream.wait_stream(main)
        with torch.cuda.stream(self._copy_stream):
            for view in self.peer_views:
                view[c, self.rank].copy_(send, non_blocking=True)
        stream = self._copy_stream

        class _Wait:
            def wait(self_inner):
                main.wait_stream(stream)
        return _Wait()

    def rows_of(self, rank):
        """Strided view ``[sizes[rank], k]``-equivalent of rank ``rank``'s rows inside the gather buffer, in their
        original order: shape ``[chunks, rows, k]`` whose flattened first two axes are the rank's rows (the last
        chunk carries ``chunks * rows - sizes[rank]`` padding rows at its end)."""
        return self.out[:, rank]

    def gathered(self):
        """All rows of all ranks in global order as ONE new tensor ``[sum(sizes), k]`` (a copy; for callers that
        want the plain layout -- the timed step never does this)."""
        parts = [self.rows_of(r).reshape(self.chunks * self.rows, self.k)[: self.sizes[r]] for r in range(self.world)]
        return torch.cat(parts, dim=0)


class ShardedProjection:
    """Convenience wrapper around a projection callable ``[b, ...] -> [b, k, 1]`` (the module itself)."""

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

    def _step(self, x_local, sizes, chunks):
        def into(x_rows, out_rows):
            out_rows.copy_(self.project_fn(x_rows).reshape(out_rows.shape))
        probe = self.project_fn(x_local[:0])
        k = probe.shape[1]
        step = ShardedStep(into, sizes, k, probe.dtype, probe.device, chunks=chunks, gather=True, group=self.group)
        step(x_local)
        return step.gathered().reshape((-1, k) + tuple(probe.shape[2:]))

    def forward_replicated(self, x_full, chunks: int = 1):
        """``x_full`` is replicated on every rank: project this rank's slice, all-gather ``y``.
        Returns the full ``[B, k, 1]`` result on every rank, rows in the original order."""
        if self.world == 1:
            return self.project_fn(x_full)
        lo, hi = shard_bounds(x_full.shape[0], self.world, self.rank)
        return self._step(x_full[lo:hi], shard_sizes(x_full.shape[0], self.world), chunks)

    def forward_gather(self, x_local, chunks: int = 1):
        """Every rank holds its own rows (equal counts or not): project them, all-gather ``y``."""
        if self.world == 1:
            return self.project_fn(x_local)
        counts = torch.zeros(self.world, dtype=torch.int64, device=x_local.device)
        counts[self.rank] = x_local.shape[0]
        dist.all_reduce(counts, group=self.group)
        return self._step(x_local, [int(c) for c in counts.tolist()], chunks)
