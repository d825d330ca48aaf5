"""World-size-2 test of the batch sharding + all-gather logic on CPU (gloo backend).

The HIP kernels cannot run here, so the per-rank compute is the CPU oracle; what is
under test is `rayen_amd.dist` (shard bounds, uneven shards, chunked gather, row order) and the
step function `bench.py --gpus N` times (`bench.make_step` / `bench.local_sizes`), driven here with a
CPU stand-in for the projection so that the code the 8-GPU run executes is the code tested.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, chunks, result_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import csd_from_cs
        from oracle import rayen_oracle as oracle
        from rayen_amd import workloads
        from rayen_amd.dist import ShardedProjection, shard_bounds

        torch.set_num_threads(1)
        cs = workloads.build_constraints(workloads.make_raw("c2", seed=3))
        buf = oracle.precompute(csd_from_cs(cs), torch.float32)

        def project(x):
            return oracle.forward(buf, x)

        gen = torch.Generator().manual_seed(17)           # same stream on every rank
        x_full = torch.empty(B, cs.n, 1).uniform_(-1, 1, generator=gen)
        sharded = ShardedProjection(project)
        y_rep = sharded.forward_replicated(x_full, chunks=chunks)
        lo, hi = shard_bounds(B, world, rank)
        y_gat = sharded.forward_gather(x_full[lo:hi], chunks=chunks)
        y_loc = sharded.forward_local(x_full[lo:hi])
        want = project(x_full)
        assert y_rep.shape == want.shape
        assert torch.equal(y_rep, want), "replicated-input path"
        assert torch.equal(y_gat, want), "local-input path"
        assert torch.equal(y_loc, want[lo:hi])
        # deliberately unequal shards on the local-input path
        cut = B // 3
        mine = x_full[:cut] if rank == 0 else x_full[cut:]
        assert torch.equal(sharded.forward_gather(mine, chunks=chunks), want)
        np.save(os.path.join(result_dir, f"ok_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,chunks", [(64, 1), (101, 1), (101, 4), (7, 3)])
def test_sharded_projection_world2(tmp_path, B, chunks):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, chunks, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert os.path.exists(tmp_path / f"ok_{rank}.npy")


def test_shard_bounds_cover_and_order():
    from rayen_amd.dist import shard_bounds, shard_sizes
    for total in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = shard_sizes(total, world)
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1


def _bench_worker(rank, world, port, scaling, chunks, config_batch, per_gpu, result_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from helpers import csd_from_cs
        from oracle import rayen_oracle as oracle
        from rayen_amd import workloads

        torch.set_num_threads(1)
        cs = workloads.build_constraints(workloads.make_raw("c2", seed=3))
        buf = oracle.precompute(csd_from_cs(cs), torch.float32)
        calls = []

        def project_into(x_rows, out_rows):                # stand-in for ops.project_raw(..., out=out_rows)
            assert out_rows.shape == (x_rows.shape[0], cs.k) and out_rows.stride(1) == 1
            calls.append(x_rows.shape[0])
            out_rows.copy_(oracle.forward(buf, x_rows)[:, :, 0])

        sizes = bench.local_sizes(config_batch, per_gpu, world, scaling)
        assert len(sizes) == world and (sum(sizes) == config_batch if scaling == "strong" else sizes == [per_gpu] * world)
        # every rank can rebuild every rank's inputs (seed + rank, as bench.py does)
        xs = [torch.empty(sizes[r], cs.n, 1).uniform_(-1, 1, generator=torch.Generator().manual_seed(1000 + r))
              for r in range(world)]
        step = bench.make_step(project_into, sizes, cs.k, torch.float32, torch.device("cpu"), gather=True, chunks=chunks)
        for _ in range(2):                                  # a step is repeatable: buffers are reused, nothing accumulates
            out = step(xs[rank])
        assert out.shape == (step.chunks, world, step.rows, cs.k)
        assert sum(calls) == 2 * sizes[rank] and max(calls) <= step.rows
        for r in range(world):
            want = oracle.forward(buf, xs[r])[:, :, 0]
            got = step.rows_of(r).reshape(-1, cs.k)[: sizes[r]]
            assert torch.equal(got, want), f"rank {rank}: rows of rank {r}"
        assert torch.equal(step.gathered(), torch.cat([oracle.forward(buf, xx)[:, :, 0] for xx in xs]))
        # the projection-only step (bench's `no_gather` leg) leaves this rank's rows in its local buffer
        local = bench.make_step(project_into, sizes, cs.k, torch.float32, torch.device("cpu"), gather=False, chunks=chunks)
        assert torch.equal(local(xs[rank]), oracle.forward(buf, xs[rank])[:, :, 0])
        np.save(os.path.join(result_dir, f"ok_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def _peer_worker(rank, world, port, chunks, sizes, result_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from helpers import csd_from_cs
        from oracle import rayen_oracle as oracle
        from rayen_amd import workloads
        from rayen_amd.dist import ShmBuffers

        torch.set_num_threads(1)
        cs = workloads.build_constraints(workloads.make_raw("c2", seed=3))
        buf = oracle.precompute(csd_from_cs(cs), torch.float32)

        def project_into(x_rows, out_rows):
            out_rows.copy_(oracle.forward(buf, x_rows)[:, :, 0])

        xs = [torch.empty(sizes[r], cs.n, 1).uniform_(-1, 1, generator=torch.Generator().manual_seed(1000 + r))
              for r in range(world)]
        rccl = bench.make_step(project_into, sizes, cs.k, torch.float32, torch.device("cpu"), gather=True, chunks=chunks)
        # the peers' buffers: files under /dev/shm that every rank maps (the host stand-in for CUDA IPC: a write into
        # views[r] lands in rank r's own tensor, as hipMemcpyPeerAsync does)
        prefix = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else result_dir, f"rayen_peer_{port}")
        peer = bench.make_step(project_into, sizes, cs.k, torch.float32, torch.device("cpu"), gather=True, chunks=chunks,
                               gather_impl="peer", peer_buffers=ShmBuffers(prefix))
        assert peer.gather_impl == "peer" and len(peer.peer_views) == world and peer.peer_views[rank] is peer.out
        for _ in range(2):                                  # repeatable: slots are overwritten, nothing accumulates
            a = rccl(xs[rank])
            b = peer(xs[rank])
        dist.barrier()
        assert a.shape == b.shape
        for r in range(world):                              # same rows in the same slots, whichever way they travelled
            assert torch.equal(peer.rows_of(r).reshape(-1, cs.k)[: sizes[r]], rccl.rows_of(r).reshape(-1, cs.k)[: sizes[r]]), (rank, r)
            # (and they are the projection of rank r's rows: the whole batch at once may sum in another order than a block)
            assert torch.allclose(peer.rows_of(r).reshape(-1, cs.k)[: sizes[r]], oracle.forward(buf, xs[r])[:, :, 0], rtol=1e-5, atol=1e-6)
        assert torch.equal(peer.gathered(), rccl.gathered())
        dist.barrier()
        if rank == 0:
            for r in range(world):
                try:
                    os.remove(f"{prefix}.{r}")
                except OSError:
                    pass
        np.save(os.path.join(result_dir, f"ok_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks,sizes", [(2, 1, [96, 96]), (2, 3, [101, 77]), (4, 2, [33, 64, 1, 50])])
def test_peer_copy_gather_fills_the_same_slots_as_the_collective(tmp_path, world, chunks, sizes):
    """`--gather-impl peer` (direct copies into the peers' gather buffers; on the GPU hipMemcpyPeerAsync through CUDA-IPC views)
    against the collective: the same rows in the same `[chunk, rank, row]` slots on every rank, equal and unequal shards,
    ragged chunks, world 2 and 4.  The copy itself is a host write into a shared mapping here; the indexing is the code the
    multi-GPU run executes."""
    port = _free_port()
    mp.spawn(_peer_worker, args=(world, port, chunks, sizes, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert os.path.exists(tmp_path / f"ok_{rank}.npy")


@pytest.mark.parametrize("scaling,chunks,config_batch,per_gpu", [("weak", 1, 0, 96), ("weak", 4, 0, 101),
                                                                 ("strong", 2, 203, 0), ("strong", 3, 64, 0)])
def test_bench_step_function_world2(tmp_path, scaling, chunks, config_batch, per_gpu):
    """`bench.py`'s multi-rank step (chunked asynchronous all-gather of y straight from the projection's output
    buffer), weak and strong sharding, ragged chunks and unequal shards, on two gloo ranks."""
    world = 2
    port = _free_port()
    mp.spawn(_bench_worker, args=(world, port, scaling, chunks, config_batch, per_gpu, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert os.path.exists(tmp_path / f"ok_{rank}.npy")


def _world4_worker(rank, world, port, result_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from helpers import csd_from_cs
        from oracle import rayen_oracle as oracle
        from rayen_amd import workloads

        torch.set_num_threads(1)
        cs = workloads.build_constraints(workloads.make_raw("c2", seed=5))
        buf = oracle.precompute(csd_from_cs(cs), torch.float32)

        def project_into(x_rows, out_rows):
            out_rows.copy_(oracle.forward(buf, x_rows)[:, :, 0])

        # (a) the config's batch sharded over four ranks with a remainder: shards of 26, 26, 25, 25 rows
        sizes = bench.local_sizes(102, 0, world, "strong")
        assert sizes == [26, 26, 25, 25]
        # (b) deliberately lopsided shards, one of them EMPTY (a rank with no rows still takes part in every collective)
        lopsided = [40, 0, 7, 13]
        reserved = []

        def set_reserve(n):                                  # stand-in for rayen_reserve_cus: set and restored per step
            reserved.append(n)
            return 0

        for shard, chunks in ((sizes, 3), (lopsided, 4), (lopsided, 1)):
            xs = [torch.empty(shard[r], cs.n, 1).uniform_(-1, 1, generator=torch.Generator().manual_seed(2000 + r))
                  for r in range(world)]
            step = bench.make_step(project_into, shard, cs.k, torch.float32, torch.device("cpu"), gather=True, chunks=chunks,
                                   reserve_cus=8, set_reserve=set_reserve)
            step(xs[rank], trace=True)
            for r in range(world):
                want = oracle.forward(buf, xs[r])[:, :, 0]
                got = step.rows_of(r).reshape(-1, cs.k)[: shard[r]]
                assert torch.equal(got, want), f"rank {rank}: rows of rank {r} ({shard}, {chunks} chunks)"
            assert torch.equal(step.gathered(), torch.cat([oracle.forward(buf, xx)[:, :, 0] for xx in xs]))
            # the per-chunk time stamps: one record per chunk, projection before its gather, chunks in order
            tr = step.last_trace
            assert len(tr) == step.chunks and sum(t["rows"] for t in tr) == shard[rank]
            assert all(t["projection_end_ms"] <= t["gather_end_ms"] for t in tr)
            assert all(tr[i]["projection_end_ms"] <= tr[i + 1]["projection_end_ms"] for i in range(len(tr) - 1))
        assert reserved == [8, 0] * 3                        # reserved for the step, handed back after it
        np.save(os.path.join(result_dir, f"ok_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_bench_step_function_world4_unequal_shards(tmp_path):
    """Four ranks, remainders, a lopsided split with an empty rank, the CU reservation handed to the projection for the
    duration of a gather step, and the per-chunk trace the first multi-GPU run will be read by."""
    world = 4
    port = _free_port()
    mp.spawn(_world4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert os.path.exists(tmp_path / f"ok_{rank}.npy")


# --------------------------------------------------------------------------- the bench entry itself, as the driver calls it
def _bench_line(args, env_extra=None, launcher=None):
    """Run ``bench.py`` (optionally under ``torch.distributed.run``) on the host backend and parse rank 0's JSON line."""
    import json
    import subprocess
    env = dict(os.environ, RAYEN_BENCH_BACKEND="gloo", RAYEN_BENCH_SETTLE_MIN_S="0.02", RAYEN_BENCH_SETTLE_MAX_S="0.2")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={launcher}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(REPO, "bench.py"), *args]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_entry_self_launches_for_gpus_2():
    """``python bench.py --gpus 2`` WITHOUT torchrun (WORLD_SIZE unset) must produce the line: it re-runs itself under
    torch.distributed.run.  Host dry run (gloo, the product's packed torch evaluator as the projection): what is under
    test is the launcher, the rank plumbing, the step and every field the first hardware run will be read by."""
    line = _bench_line(["--gpus", "2", "--config", "c2", "--batch", "200", "--steps", "3", "--warmup", "1", "--chunks", "2"])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"]["name"] == "gloo"
    assert line["config"]["global_batch"] == 400 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["steps"] == 3 and line["warmup"] == 1
    assert line["no_gather"]["scaling_efficiency"] > 0 and line["no_gather"]["solo_rank0_ms_per_step"] > 0
    g = line["gather"]
    assert g["bytes_received_per_rank"] == 200 * 16 * 4 and g["GBps_per_rank"] > 0
    assert g["xgmi_peak_GBps_per_rank"] == pytest.approx(76.8) and 0 < g["xgmi_frac"] < 1
    assert len(g["per_chunk_rank0"]) == 2
    assert line["settle"]["windows"] >= 2 and line["first_window_ms"] > 0 and line["settled_ms"] > 0
    assert line["violations_gt_1e-6"] == 0 and line["violations_checked_rows"] == 200
    assert "DRY RUN" in line["config"]["kernel"]


def test_bench_entry_under_the_launcher_with_one_rank_is_the_plain_run():
    """N = 1 through ``torch.distributed.run`` and N = 1 as a plain process describe the same job (the values
    themselves are host timings here)."""
    args = ["--gpus", "1", "--config", "c2", "--batch", "128", "--steps", "2", "--warmup", "1"]
    plain = _bench_line(args)
    launched = _bench_line(args, launcher=1)
    for key in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "config"):
        assert plain[key] == launched[key], key
    assert "rccl_ranks" not in plain and "gather" not in plain and "gather" not in launched


def test_bench_entry_strong_scaling_shards_the_config_batch():
    line = _bench_line(["--gpus", "2", "--config", "c1", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--no-gather"])
    assert line["config"]["global_batch"] == 500 and line["config"]["batch_per_gpu"] == 250
    assert "gather" not in line and line["no_gather"]["value"] > 0
