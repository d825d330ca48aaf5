#!/usr/bin/env python
"""Throughput of the RAYEN projection on MI355X -- the driver's bench contract.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5]

A "step" is one pass of the hot path (``ConstraintModule.forward`` with the identity
mapper = one launch of the fused projection) over one batch of synthetic directions
that is already resident in HBM.  Default workload: BASELINE.json ``configs[2]``, the
headline -- k=64, 128 linear + 4 quadratic + 2 SOC constraints, batch 262144 per GPU,
fp32.  The other configs are selected with ``--config`` (their lines are committed
under ``profiles/bench/``).

N>1 runs one process per GPU (torchrun, backend nccl = RCCL).  The batch dimension is
sharded -- fixed per-GPU work by default (weak scaling), ``--scaling strong`` shards the
config's own batch (config 5: 2M rows over the ranks) -- and the step is the north_star's:
every rank projects its rows and ONE all-gather of ``y`` (chunked, asynchronous, issued
while the next chunk is projected: ``rayen_amd.dist.ShardedStep``) leaves every output on
every rank.  ``value`` is that step; the same line carries ``no_gather`` (the projection
alone, what a data-parallel training step needs).  ``--no-gather`` times only the latter.

One JSON line on rank 0: metric / value (whole-job projections/s) plus ``roofline``
(dominant kernel, live HIP-event timing on the launch stream, PMC traffic from the
committed rocprofv3 summary of the same command) and ``cpu_baseline`` (the PyTorch-CPU
oracle, same workload, timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector = FP32 MFMA peak (spec)
PEAK_BF16_TFLOPS = 2516.8  # dense bf16 MFMA = 16 x the fp32 MFMA rate (same guide; 2495 measured)
PEAK_FP64_TFLOPS = 78.6    # MI355X datasheet: FP64 vector = FP64 matrix (v_mfma_f64_16x16x4_f64)
PEAK_HBM_GBS = 8000.0      # HBM3E spec (≈6.3 TB/s achievable)
GRAPH_STEPS = 50          # steps captured per HIP graph when the step is launch-bound
SETTLE_LAUNCHES = 150      # untimed launches of a comparison family before its own timing
# Clocks of a fresh box: the device needs an unknown stretch of load before its clocks stop moving (round 3: a fixed
# 120 ms settle read 0.0646 ms on the driver's box where a warm device reads 0.060).  The settle is ADAPTIVE: untimed
# windows of the invocation's own shape (W + K steps) are repeated until two consecutive windows agree within
# SETTLE_TOL -- at least SETTLE_MIN_S of load, at most SETTLE_MAX_S -- and the line reports the first and the last
# window next to the official W + K measurement that follows.
SETTLE_TOL = 0.02
SETTLE_MIN_S = float(os.environ.get("RAYEN_BENCH_SETTLE_MIN_S", "0.25"))
SETTLE_MAX_S = float(os.environ.get("RAYEN_BENCH_SETTLE_MAX_S", "2.0"))
XGMI_GBPS_PER_LINK_PER_DIRECTION = 76.8   # MI355X: 7 links x 153.6 GB/s bidirectional per GPU, one direct link per peer
BACKEND = os.environ.get("RAYEN_BENCH_BACKEND", "nccl")   # "gloo": CPU dry run of exactly this entry (tests/test_dist_gloo.py)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c4", "c5", "c5r"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = the BASELINE.json size)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank projects the config's per-GPU batch | strong: the config's batch is sharded")
    ap.add_argument("--no-gather", action="store_true", help="N>1: time the projection alone (no all-gather of y)")
    ap.add_argument("--gather", action="store_true",
                    help="with --force-dist on one GPU: run the all-gather step anyway (one-rank RCCL group)")
    ap.add_argument("--gather-impl", choices=["rccl", "peer"], default="rccl",
                    help="the all-gather of y: RCCL's collective, or direct copies into the peers' buffers (hipMemcpyPeerAsync / SDMA, "
                         "no compute units; rayen_amd/dist.py) -- an A/B for multi-GPU runs")
    ap.add_argument("--chunks", type=int, default=2, help="row blocks per rank whose all-gathers overlap the next block's projection")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="compute units the projection's persistent grid leaves to RCCL's kernels during the gather step")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a HIP graph (auto: launch-bound batches, B*k < 2^20)")
    ap.add_argument("--no-rotate", action="store_true",
                    help="time ONE (x, y) pair (cache-resident at config 3) instead of a rotation larger than the Infinity Cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true",
                    help="skip the comparison runs on the other fp32 kernel families (profiling: one dominant kernel)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank (exercises the multi-GPU code path)")
    ap.add_argument("--mapper", type=int, default=0, metavar="D",
                    help="put the module's nn.Linear(D, n) mapper in front (create_map=True); 0 = identity mapper")
    ap.add_argument("--no-fuse", action="store_true", help="with --mapper: run the mapper as its own GEMM")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


def _note(msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(raw, cs, B, dtype, budget_s, rng=1.0):
    """Reference op sequence (oracle/rayen_oracle.py) on the host cores, bounded sample of the workload."""
    from oracle import rayen_oracle as oracle
    cores = os.cpu_count() or 1
    csd = {key: getattr(cs, key) for key in ("A_p", "b_p", "NA_E", "yp", "z0", "y0")}
    csd.update(P=raw["P"], q=raw["q"], r=raw["r"], M=raw["M"], s=raw["s"], c=raw["c"], d=raw["d"],
               F=raw["F"])
    buf = oracle.precompute(csd, dtype)
    gen = torch.Generator().manual_seed(1234)
    Bs = min(B, 32768)
    x = torch.empty(Bs, cs.n, 1, dtype=dtype).uniform_(-rng, rng, generator=gen)

    nan_rows = [0]

    def timed(xx):
        t0 = time.perf_counter()
        y_cpu = oracle.forward(buf, xx, check_nan=False)     # (the reference's NaN assert, CM:531, is not part of the timing)
        dt = time.perf_counter() - t0
        nan_rows[0] = int(torch.isnan(y_cpu).any(dim=1).sum())
        return dt

    with torch.no_grad():
        # PyTorch-CPU does not scale to every core on this op mix: probe a few thread counts, keep the best
        probe = x[:4096]
        rates = {}
        # (LMI sets: batched eigvalsh of 20 x 20 matrices does not scale past a few threads and collapses beyond)
        candidates = (1, 4, 8, 16) if len(raw["F"]) else (1, 4, 8, 16, 32, 64, cores)
        for threads in sorted({t for t in candidates if t <= cores}):
            torch.set_num_threads(threads)
            _note(f"cpu baseline probe, {threads} threads")
            first = timed(probe)
            rates[threads] = probe.shape[0] / min(first, timed(probe), timed(probe))
            if first > 2.0:        # batched eigvalsh of small matrices collapses with many threads: stop probing upwards
                break
        threads = max(rates, key=rates.get)
        torch.set_num_threads(threads)
        timed(x)
        best, reps, t_start = float("inf"), 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t_start < budget_s and reps < 30):
            best = min(best, timed(x))
            reps += 1
    return {"value": Bs / best, "unit": "projections/s", "cores": threads, "kind": "port",
            "host_cores": cores,
            # (config 5: the reference's own op sequence takes sqrt of slightly negative radicands at fp32 on the
            # corridor set -- NaN outputs, which its assert at CM:531 would stop on; the HIP path is finite there)
            "reference_nan_rows_in_sample": nan_rows[0],
            "sample": f"B={Bs} slice of the same workload, best of {reps} calls after warm-up, "
                      f"{str(dtype).split('.')[-1]}, torch {torch.__version__} CPU, {threads} threads "
                      f"(best of thread counts {sorted(rates)})"}


def make_step(project_into, sizes, k, dtype, device, gather, chunks, group=None, gather_alone=False,
              reserve_cus=0, set_reserve=None, gather_impl="rccl", peer_buffers=None):
    """The multi-rank step ``bench.py`` times: ``rayen_amd.dist.ShardedStep`` around ``project_into(x_rows,
    out_rows)`` -- on the GPU the C-ABI projection writing straight into the gather's send buffer; in
    ``tests/test_dist_gloo.py`` a CPU stand-in, so the code the 8-GPU driver run executes is the code tested."""
    from rayen_amd.dist import ShardedStep
    return ShardedStep(project_into, sizes, k, dtype, device, chunks=chunks, gather=gather, group=group,
                       gather_alone=gather_alone, reserve_cus=reserve_cus, set_reserve=set_reserve,
                       gather_impl=gather_impl, peer_buffers=peer_buffers)


def local_sizes(config_batch, per_gpu_batch, world, scaling):
    """Rows per rank: ``weak`` = the per-GPU batch on every rank, ``strong`` = the config's batch sharded."""
    from rayen_amd.dist import shard_sizes
    if scaling == "strong":
        return shard_sizes(config_batch, world)
    return [per_gpu_batch] * world


def _sync(on_gpu):
    if on_gpu:
        torch.cuda.synchronize()


def timed_loop(step, x, steps, warmup, use_dist, graph=False, on_gpu=True):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; returns (wall seconds,
    device ms per step from HIP events recorded on the launch stream; on the CPU dry run: the wall time)."""
    with torch.no_grad():
        replay, per = step, 1
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step(x)
            torch.cuda.current_stream().wait_stream(side)
            # launch-bound batches: GRAPH_STEPS steps per graph (one graph launch amortised over them)
            per = max(1, min(GRAPH_STEPS, steps))
            while steps % per:
                per -= 1
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(per):
                    step(x)
            replay = lambda _x: g.replay()          # noqa: E731
        for _ in range(-(-warmup // per)):
            replay(x)
        _sync(on_gpu)
        if use_dist:
            dist.barrier()
        _sync(on_gpu)
        if on_gpu:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if on_gpu:
            ev0.record()
        for _ in range(steps // per):
            replay(x)
        if on_gpu:
            ev1.record()
        _sync(on_gpu)
        if use_dist:
            dist.barrier()
        _sync(on_gpu)
        elapsed = time.perf_counter() - t0
    return elapsed, (ev0.elapsed_time(ev1) if on_gpu else elapsed * 1e3) / steps


def settle(step, x, steps, warmup, graph, on_gpu):
    """Untimed windows of W + K steps until two consecutive windows agree within SETTLE_TOL (clocks of a fresh box);
    returns {"first_window_ms", "settled_ms", "windows", "seconds", "converged"} -- per-step device times."""
    t_start = time.perf_counter()
    history = []
    while True:
        _, ms = timed_loop(step, x, steps, warmup, False, graph=graph, on_gpu=on_gpu)
        history.append(ms)
        spent = time.perf_counter() - t_start
        agree = len(history) >= 2 and abs(history[-1] - history[-2]) <= SETTLE_TOL * history[-1]
        if (agree and spent >= SETTLE_MIN_S) or spent >= SETTLE_MAX_S:
            return {"first_window_ms": history[0], "settled_ms": history[-1], "windows": len(history),
                    "seconds": spent, "converged": bool(agree),
                    "what": f"untimed windows of W+K steps until two in a row agree within {SETTLE_TOL:.0%} "
                            f"(at least {SETTLE_MIN_S:g} s, at most {SETTLE_MAX_S:g} s); the W+K measurement follows"}


L3_BYTES = 256 << 20       # MI355X Infinity Cache (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE count its hits as traffic
ROTATE_FOOTPRINT = 2 * L3_BYTES   # the timed loop cycles through (x, y) pairs of at least this many bytes in total
ROTATE_MAX_PAIRS = 16


class RotatingStep:
    """The step over ``pairs`` distinct input batches, the last ``pairs`` outputs kept alive: launch i reads x[i % pairs] and
    writes a y no other launch of the cycle touches, so a cyclic footprint of >= 2 x the 256 MiB Infinity Cache passes
    through HBM on every launch (a single (x, y) pair of config 3 is 128 MiB and would live in that cache: round-5 verdict,
    weak 5).  Same module call, same kernel, same values per batch."""

    def __init__(self, step, xs):
        self.step, self.xs, self.ring, self.i = step, xs, [None] * len(xs), 0

    def __call__(self, _x=None):
        i = self.i
        self.ring[i] = None                        # (this slot's block goes back to the allocator before the call takes one)
        self.ring[i] = self.step(self.xs[i])
        self.i = i + 1 if i + 1 < len(self.xs) else 0
        return self.ring[i]


def rotation_pairs(pair_bytes):
    """(x, y) pairs the timed loop cycles through: enough for ROTATE_FOOTPRINT; 1 (no rotation) where that would take more
    than ROTATE_MAX_PAIRS -- the launch-bound configs, whose whole working set is a few hundred KiB either way."""
    need = -(-ROTATE_FOOTPRINT // max(pair_bytes, 1))
    return need if 2 <= need <= ROTATE_MAX_PAIRS else (1 if need > ROTATE_MAX_PAIRS else 2)


def self_launch(argv, gpus):
    """``python bench.py --gpus N`` without torchrun: re-run this file under ``torch.distributed.run`` (one process per
    GPU on this node, rendezvous on 127.0.0.1) and hand its exit code back.  Rank 0's JSON line is the child's stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    _note(f"WORLD_SIZE is not set: launching {gpus} ranks -- {' '.join(cmd[1:8])} bench.py ...")
    return subprocess.call(cmd, env=env)


def profiled_traffic(config, dtype_tag, batch, kernel_tag):
    """HBM bytes per launch of this workload's dominant kernel from the newest committed rocprofv3 PMC summary
    (``profiles/r*_<config>_*rocprofv3.json``, written by scripts/summarize_profile.py from separate ``--pmc``
    passes of this same command; FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE).  ``bench.py`` itself cannot
    collect PMC counters, so a workload that was never profiled reports None."""
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{config}_*rocprofv3.json"))):
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        line = prof.get("bench_line_under_profiler", {})
        cfg = line.get("config", {})
        if line.get("dtype") != dtype_tag or cfg.get("batch_per_gpu") != batch or cfg.get("kernel") != kernel_tag:
            continue
        traffic = prof.get("derived", {}).get("hbm_bytes_per_launch")
        if traffic is not None:
            best = (traffic, os.path.relpath(path, REPO))
    return best


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(argv, args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        _note(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs")
    on_gpu = BACKEND != "gloo"
    # stdout carries ONE line, the JSON: whatever libraries write to file descriptor 1 on the way (RCCL's version banner comes
    # through C stdio at the first collective) goes to stderr instead
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if on_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")             # dry run of the launcher and the step on the host (not a measurement)
        torch.set_num_threads(1)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from rayen_amd import ops, workloads
    from rayen_amd.constraint_module import ConstraintModule

    dtype = torch.float32 if args.dtype == "fp32" else torch.float64
    torch.set_default_dtype(dtype)
    t_setup = time.perf_counter()
    raw = workloads.make_raw(args.config, seed=0)
    cs = workloads.build_constraints(raw)
    if args.mapper:
        torch.manual_seed(0)
        layer = ConstraintModule(cs, input_dim=args.mapper, method="RAYEN", create_map=True).to(device)
        layer.fuse_mapper = not args.no_fuse
    else:
        layer = ConstraintModule(cs, method="RAYEN", create_map=False).to(device)
    layer.check_nan = False                      # no host sync inside the timed region
    t_module = time.perf_counter() - t_setup
    config_batch = workloads.CONFIGS[args.config][2]
    per_gpu = args.batch or (config_batch // 8 if args.config in ("c5", "c5r") else config_batch)   # c5: 2M = 8 x 262144
    sizes = local_sizes(args.batch * world if args.batch else config_batch, per_gpu, world, args.scaling)
    B = sizes[rank]
    total_rows = sum(sizes)
    rng = workloads.CONFIGS[args.config][3]
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    x = torch.empty(B, args.mapper or cs.n, 1, device=device, dtype=dtype).uniform_(-rng, rng, generator=gen)
    t_pack = time.perf_counter()
    dp = layer.device_pack(device)[0] if on_gpu else None
    _sync(on_gpu)
    t_pack = time.perf_counter() - t_pack
    gather = (world > 1 or (args.gather and use_dist)) and not args.no_gather and not args.mapper
    graph = on_gpu and (args.graph == "on" or (args.graph == "auto" and B * cs.k < (1 << 20) and not gather))

    def module_step(xx):
        return layer(xx)

    last = {}

    if on_gpu:
        def project_into(x_rows, out_rows):
            ops.project_raw(x_rows.reshape(x_rows.shape[0], -1), dp, want_active=False, want_kappa=False, out=out_rows)
        from rayen_amd import _lib as _rlib
        set_reserve = _rlib.load().rayen_reserve_cus
    else:
        def project_into(x_rows, out_rows):       # the product's host evaluator (rayen_amd/eager.py) stands in
            out_rows.copy_(layer(x_rows)[:, :, 0])
        set_reserve = None

    def gather_step(project):
        return make_step(project, sizes, cs.k, dtype, device, gather=True, chunks=args.chunks, gather_alone=True,
                         reserve_cus=args.reserve_cus, set_reserve=set_reserve,
                         gather_impl=args.gather_impl if on_gpu else "rccl")

    sharded = gather_step(project_into) if gather else None

    # the timed step cycles through enough distinct (x, y) pairs to defeat the 256 MiB Infinity Cache (RotatingStep)
    elem = 4 if dtype == torch.float32 else 8
    pairs = rotation_pairs(B * (x.shape[1] + cs.k) * elem) if (on_gpu and not graph and B and not args.no_rotate) else 1
    single_step = module_step
    if pairs > 1:
        xs = [x] + [torch.empty_like(x).uniform_(-rng, rng, generator=gen) for _ in range(pairs - 1)]
        module_step = RotatingStep(single_step, xs)
    _note(f"{args.config} {args.dtype} B={B} per GPU, world {world}, graph={graph}, gather={gather}, backend={BACKEND}, "
          f"{pairs} (x, y) pair(s) in rotation")
    settled = settle(module_step, x, args.steps, args.warmup, graph, on_gpu)
    _note(f"settled after {settled['windows']} windows / {settled['seconds']:.2f} s "
          f"(first {settled['first_window_ms']:.4f} ms, last {settled['settled_ms']:.4f} ms); timing the projection")
    # ---- rank 0 alone (the others wait at the barrier behind it): what one GPU does with nobody else on the node busy
    solo_ms = None
    if world > 1:
        if rank == 0:
            _, solo_ms = timed_loop(module_step, x, args.steps, args.warmup, False, graph=graph, on_gpu=on_gpu)
        dist.barrier()
    # ---- the projection alone (the whole step at N = 1)
    elapsed_p, dev_ms = timed_loop(module_step, x, args.steps, args.warmup, use_dist, graph=graph, on_gpu=on_gpu)
    if on_gpu:
        from rayen_amd import _lib as _klib
        timed_kernel = _klib.load().rayen_last_forward_kernel()      # which instruction stream served the TIMED launches
    # ---- the same launches on ONE (x, y) pair (128 MiB at config 3: resident in the Infinity Cache) -- reported beside
    l3_ms = None
    if pairs > 1:
        module_step.ring = [None] * pairs
        _, l3_ms = timed_loop(single_step, x, args.steps, args.warmup, False, graph=graph, on_gpu=on_gpu)
    with torch.no_grad():
        y = single_step(x)
    # ---- projection + all-gather of y (the north_star's multi-GPU step)
    elapsed_g = None
    if gather:
        elapsed_g, dev_ms_g = timed_loop(sharded, x, args.steps, args.warmup, use_dist, on_gpu=on_gpu)
        got = sharded.rows_of(rank).reshape(-1, cs.k)[:B]
        assert torch.equal(got, y[:, :, 0]), "gathered rows differ from the local projection"
        last["dev_ms_gather"] = dev_ms_g
        sharded(x, trace=True)                                     # one more step, time-stamped per chunk (untimed)
        last["gather_trace"] = sharded.last_trace
        # the collective alone, same buffers and chunks, nothing projected: what the links do with the CUs idle
        alone = gather_step(lambda x_rows, out_rows: None)
        elapsed_a, _ = timed_loop(alone, x, args.steps, args.warmup, use_dist, on_gpu=on_gpu)
        last["gather_alone_s"] = elapsed_a
        del alone

    ranks_seen = 1
    t = torch.tensor([elapsed_p, dev_ms, elapsed_g or 0.0, last.get("dev_ms_gather", 0.0), last.get("gather_alone_s", 0.0)],
                     device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, device=device, dtype=torch.float64)
        dist.all_reduce(ones)                    # every rank of the group adds one: the number of ranks the backend reached
        ranks_seen = int(round(float(ones)))
    elapsed_p, dev_ms, elapsed_g, dev_ms_g, elapsed_a = (float(v) for v in t)
    elapsed = elapsed_g if gather else elapsed_p

    _note("timed; checking feasibility")
    # feasibility of what was just computed (fp64 residuals on a slice, outside the timed region)
    sl = y[: min(B, 65536), :, 0].double().cpu().numpy()
    row_violation = cs.getViolationRows(sl) if B else sl[:, 0]
    max_violation = float(row_violation.max()) if B else 0.0
    violation_detail = workloads.violation_report(cs, sl[:8192]) if B else None

    # the same rows with the inward bias on (RAYEN_PREPARE_INWARD_BIAS, include/rayen_hip.h: the fp32 images evaluate
    # (1 + eps) kappa, clipped samples stop eps short of the boundary) -- feasibility before / after and what it costs in parity
    bias_report = None
    if on_gpu and world == 1 and rank == 0 and not args.mapper and dtype == torch.float32 and B and not args.no_families:
        bias_report = {"what": "fp64 residuals of the fp32 outputs on the same rows, pack built with RAYEN_INWARD_BIAS=eps (0 = the "
                               "reference's step; the module's default is 2^-20); shift = per-row inf-norm distance from the eps = 0 "
                               "outputs, relative to the row",
                       "rows": int(sl.shape[0]), "module_default": "2^-20" if getattr(layer, "inward_bias", False) else "0"}
        import numpy as _np
        plain = None
        for log2 in (None, -22, -20, -19):
            os.environ["RAYEN_INWARD_BIAS"] = repr(0.0 if log2 is None else 2.0 ** log2)
            try:
                biased = ConstraintModule(cs, method="RAYEN", create_map=False).to(device)
                biased.check_nan = False
                with torch.no_grad():
                    yb = biased(x[: sl.shape[0]])[:, :, 0].double().cpu().numpy()
            finally:
                del os.environ["RAYEN_INWARD_BIAS"]
            if plain is None:
                plain = yb
            rows_b = cs.getViolationRows(yb)
            shift = _np.abs(yb - plain).max(axis=1) / _np.maximum(_np.abs(plain).max(axis=1), 1e-30)
            bias_report["0" if log2 is None else f"2^{log2}"] = {
                "max_violation": float(rows_b.max()), "violations_gt_0": int((rows_b > 0).sum()),
                "violations_gt_1e-6": int((rows_b > 1e-6).sum()), "max_rel_shift": float(shift.max())}
            del biased

    if rank == 0:
        bytes_pp, flops_pp = workloads.algorithmic_work(cs)
        if args.mapper:
            bytes_pp += 4 * (args.mapper - cs.n)
            flops_pp += 2 * args.mapper * cs.n
        if dtype == torch.float64:
            bytes_pp *= 2
        kern_s = dev_ms * 1e-3
        tflops = flops_pp * B / kern_s / 1e12
        gbs = bytes_pp * B / kern_s / 1e9
        ai = flops_pp / bytes_pp
        lmi = cs.has_lmi_constraints
        from rayen_amd import _lib
        if on_gpu:
            info = dp.info()
            split = dtype == torch.float32 and info.mfma_f32 in (2, 3)
            pieces = {2: 6.0, 3: 3.0}.get(info.mfma_f32, 1.0)    # piece products per fp32 product
            kernel_tag = (({3: "mfma_pair_f16x2 (fp32-grade)", 2: "mfma_split_bf16x3 (fp32-grade)", 1: "mfma_f32"}.get(info.mfma_f32, "lmi_lanes" if lmi else "generic"))
                          if dtype == torch.float32 else ("mfma_f64" if info.mfma_f64 else ("lmi_lanes" if lmi else "generic")))
            served_by = timed_kernel      # (read right behind the timed loop: the checks since then launched smaller batches)
            if served_by == _lib.KERNEL_PAIR_IO:
                kernel_tag = "mfma_pair_io_f16x2 (fp32-grade; rows of v and y trickled through LDS under the tile walk)"
            elif served_by == getattr(_lib, "KERNEL_PAIR_WL", -1):
                kernel_tag = "mfma_pair_wl_f16x2 (fp32-grade; the image of W resident in LDS, rows straight from / to memory, four waves per SIMD)"
            elif served_by == getattr(_lib, "KERNEL_PAIR_WS", -1):
                kernel_tag = "mfma_pair_ws_f16x2 (fp32-grade; W resident in the registers of a workgroup's waves, batch streamed through LDS)"
        else:
            info, split, pieces, served_by = None, False, 1.0, -1
            kernel_tag = "HOST DRY RUN of the launcher and the step (RAYEN_BENCH_BACKEND=gloo): packed torch evaluator, not a measurement"
        # The split-operand kernels rebuild every fp32 product from three f16 (pairs) or six bf16 (triples) MFMA products
        # (fp32-grade results), so their matrix ceiling in ALGORITHMIC fp32 flops is the dense 16-bit peak / 3 or / 6,
        # not the fp32 MFMA peak
        peak_tf = (PEAK_BF16_TFLOPS / pieces if split else PEAK_FP32_TFLOPS) if dtype == torch.float32 else PEAK_FP64_TFLOPS
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        if ai > ridge:
            roof = {"bound": "mfma", "achieved": tflops, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": tflops / peak_tf, "traffic": None}
            if split:
                roof["peak_basis"] = "dense f16/bf16 MFMA peak %.1f / %d piece products per fp32 product" % (PEAK_BF16_TFLOPS, pieces)
                roof["frac_of_fp32_mfma_peak"] = tflops / PEAK_FP32_TFLOPS
            elif lmi:
                roof["bound"] = "valu"
                roof["peak_basis"] = ("fp32/fp64 vector ALU peak (numerically the MFMA peak of the dtype): the per-sample "
                                      "eigen-solve (Householder + Sturm) has no matrix-core form")
        else:
            roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": gbs / PEAK_HBM_GBS, "traffic": None}
        dtype_tag = "f32" if dtype == torch.float32 else "f64"
        found = None if args.mapper else profiled_traffic(args.config, dtype_tag, B, kernel_tag)
        if found:
            roof["traffic"], roof["traffic_source"] = found
            roof["traffic_unit"] = "bytes/launch (algorithmic: %d)" % (bytes_pp * B)
        roof.update({"kernel_ms": dev_ms, "algorithmic_flops_per_projection": flops_pp,
                     "algorithmic_bytes_per_projection": bytes_pp, "hbm_GBps": gbs,
                     "hbm_frac": gbs / PEAK_HBM_GBS, "TFLOPs": tflops,
                     "hip_graph_replay": (f"{GRAPH_STEPS} steps per graph" if graph else False)})
        pair_bytes = B * ((args.mapper or cs.n) + cs.k) * elem
        roof["footprint"] = {"pairs_in_rotation": pairs, "bytes": pairs * pair_bytes, "infinity_cache_bytes": L3_BYTES,
                             "beyond_infinity_cache": bool(pairs * pair_bytes >= 2 * L3_BYTES),
                             "what": "the timed loop cycles through this many distinct (x, y) batches; hbm_GBps is an HBM "
                                     "figure only when the cycle is at least twice the Infinity Cache"}
        if l3_ms is not None:
            roof["l3_resident"] = {"kernel_ms": l3_ms, "hbm_GBps": bytes_pp * B / (l3_ms * 1e-3) / 1e9,
                                   "value": B / (l3_ms * 1e-3), "TFLOPs": flops_pp * B / (l3_ms * 1e-3) / 1e12,
                                   "what": "same launches on ONE (x, y) pair (cache-resident where it fits 256 MiB): NOT an HBM figure"}
        out = {
            "metric": "feasible projections/sec at k=64, 128 lin+4 quad+2 SOC; max violation"
                      if args.config == "c3" else f"feasible projections/sec ({args.config})",
            "value": total_rows * args.steps / elapsed,
            "unit": "projections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype_tag, "data": "synthetic",
            "config": {"workload": f"{args.config}: k={cs.k} n={cs.n}, {cs.A_p.shape[0]} linear + "
                                   f"{len(cs.qcs)} quadratic + {len(cs.socs)} SOC"
                                   f"{' + 1 LMI' if cs.has_lmi_constraints else ''}, "
                                   f"batch {B} per GPU, v~U(-{rng:g},{rng:g})"
                                   + (" [the corridor trajectory set restated from the reference's MATLAB generator -- "
                                      "clamped cubic B-spline, 12 intervals, 6 hull regions, Bezier points of every interval "
                                      "inside its region, 15 boundary equalities, 72 rank-3 limits; corridor_dim3.mat itself "
                                      "is an absent LFS pointer: rayen_amd/workloads.py::corridor_spline]"
                                      if args.config == "c5" else
                                      " [random 288-row stand-in of rounds 1-2 with the corridor set's counts: 15 equalities, "
                                      "72 rank-3 quadratics]" if args.config == "c5r" else ""),
                       "batch_per_gpu": B, "global_batch": total_rows,
                       "parallelism": f"batch-sharded x{world}" + (f" + all-gather(y) in {sharded.chunks} chunks" if gather else ""),
                       "kernel": kernel_tag},
            "max_violation": max_violation,
            "violations_gt_1e-6": int((row_violation > 1e-6).sum()) if B else 0,
            "violations_gt_0": int((row_violation > 0).sum()) if B else 0,
            "violations_checked_rows": int(sl.shape[0]),
            "settle": settled,
            "first_window_ms": settled["first_window_ms"], "settled_ms": settled["settled_ms"],
            "kernel_ms": dev_ms,
            "setup": {"module_s": t_module, "pack_create_s": t_pack,
                      "what": "ConvexConstraints + ConstraintModule construction | rayen_pack_create (device images + "
                              "the creation-time measurement of the fp32 families), once per module and device"},
            # (per family: the residual next to what rounding a feasible y to fp32 alone can leave -- sets with large
            # coefficients, configs 5 / 5r, sit above 1e-6 in absolute terms at a ratio of a few units)
            "violation_detail": violation_detail,
            "inward_bias": bias_report,
            "roofline": roof,
        }
        if use_dist:
            out["rccl_ranks"] = ranks_seen
            out["backend"] = {"name": dist.get_backend() if dist.is_initialized() else BACKEND, "world_size": world,
                              "what": "rccl_ranks = sum over ranks of 1 through an all-reduce on this backend"}
        if world > 1 or gather:
            out["no_gather"] = {"value": total_rows * args.steps / elapsed_p, "unit": "projections/s",
                                "ms_per_step": elapsed_p / args.steps * 1e3,
                                "what": "the projection alone on every rank (no collective), same inputs and step count"}
            out["value_no_gather"] = out["no_gather"]["value"]        # (top-level siblings for one-level parsers)
            if solo_ms is not None:
                # rank 0 alone on the node at the same per-rank rows, against all ranks at once (max over ranks)
                solo_rate = sizes[0] / (solo_ms * 1e-3)
                out["no_gather"].update({"solo_rank0_ms_per_step": solo_ms, "all_ranks_device_ms_per_step": dev_ms,
                                         "scaling_efficiency": (total_rows / (dev_ms * 1e-3)) / (world * solo_rate),
                                         "scaling_efficiency_basis": "device time: (rows of all ranks / slowest rank's "
                                                                     "step) / (N x rank 0's rate with the other ranks idle)"})
                out["scaling_efficiency"] = out["no_gather"]["scaling_efficiency"]
                out["scaling_efficiency_of"] = "value_no_gather (the projection alone; `value` includes the all-gather of y)"
            if gather:
                recv = (total_rows - B) * cs.k * (4 if dtype == torch.float32 else 8)
                peers = max(world - 1, 1)
                link_peak = peers * XGMI_GBPS_PER_LINK_PER_DIRECTION
                ga_s = elapsed_a / args.steps if elapsed_a else None
                gbps = recv / ga_s / 1e9 if ga_s else None
                out["gather"] = {"bytes_received_per_rank": recv,
                                 "alone_ms_per_step": ga_s * 1e3 if ga_s else None,
                                 "GBps_per_rank": gbps,
                                 "xgmi_peak_GBps_per_rank": link_peak if world > 1 else None,
                                 "xgmi_frac": (gbps / link_peak) if (gbps and world > 1) else None,
                                 "xgmi_basis": f"{peers} direct links x {XGMI_GBPS_PER_LINK_PER_DIRECTION} GB/s inbound "
                                               "(153.6 GB/s per link, both directions); the collective alone on the "
                                               "step's buffers, nothing projected",
                                 "step_minus_projection_ms": (elapsed_g - elapsed_p) / args.steps * 1e3,
                                 "chunks": sharded.chunks, "device_ms_per_step": dev_ms_g,
                                 "reserved_cus": args.reserve_cus,
                                 "per_chunk_rank0": last.get("gather_trace"),
                                 "how_to_read": "overlap happened if chunk c's gather_end_ms is not behind chunk c+1's "
                                                "projection_end_ms by the gather's own duration"}
        if on_gpu and world == 1 and not args.mapper and not args.no_families and B:
            # the training step of the same workload: forward with the arg-max record, then the backward (the two launches
            # `loss.backward()` through the layer costs, examples/main.py:166-171), timed the same way, outside `value`
            g_in = torch.empty(B, cs.k, device=device, dtype=dtype).uniform_(-1, 1)
            xv = x[:, :cs.n, 0].contiguous() if x.dim() == 3 else x[:, :cs.n].contiguous()
            with torch.no_grad():
                _, k_rec, a_rec = ops.project_raw(xv, dp, want_active=True)
                # (best of three loops each: these calls allocate their outputs, and a loop that meets the caching
                # allocator growing its pool reads several times too slow)
                ms_ft = min(timed_loop(lambda t_: ops.project_raw(t_, dp, want_active=True), xv, args.steps, args.warmup, False)[1]
                            for _ in range(3))
                ms_b = min(timed_loop(lambda t_: ops.backward_raw(t_, k_rec, a_rec, g_in, dp), xv, args.steps, args.warmup, False)[1]
                           for _ in range(3))
            bwd_names = {0: "lane-per-sample", 1: "exact-fp32 MFMA (dense forms)", 2: "exact-fp32 MFMA (general shapes)",
                         3: "f16 pairs (packed low-rank forms)", 4: "four lanes per sample (LMI)", 5: "one wave per sample (LMI)",
                         7: "f16 pairs, dense forms resident in LDS, one launch (round 6)"}
            out["training_step"] = {
                "forward_with_record_ms": ms_ft, "backward_ms": ms_b, "unit": "ms per call, eager, HIP events",
                "backward_kernel": bwd_names.get(int(info.bwd_f32), str(info.bwd_f32)) if dtype == torch.float32 else "fp64 kernels",
                "backward_check_pair_vs_fp64": info.bwd32_check_pair, "backward_check_exact_vs_fp64": info.bwd32_check_exact,
                "clipped_fraction": float((k_rec > 1).double().mean())}
        if on_gpu and world == 1 and not args.mapper and not args.no_families and B and dtype == torch.float32 \
                and cs.n <= 64 and cs.n % 4 == 0 and dp.mapper_fusable(cs.n):
            # the same workload behind the module's own mapper (create_map=True, the reference's default: rayen/constraint_module.py
            # :259-263, 525): x -> v = Wm x + b -> y in ONE launch, timed the same way, outside `value`
            try:
                torch.manual_seed(0)
                mapped = ConstraintModule(cs, input_dim=cs.n, method="RAYEN", create_map=True).to(device)
                mapped.check_nan = False
                xm = torch.empty(B, cs.n, device=device, dtype=dtype).uniform_(-1, 1)
                with torch.no_grad():
                    for _ in range(2 * SETTLE_LAUNCHES):
                        mapped(xm)
                torch.cuda.synchronize()
                ms_m = min(timed_loop(lambda xx: mapped(xx), xm, args.steps, args.warmup, False, graph=graph)[1] for _ in range(3))
                out["module_with_mapper"] = {
                    "value": B / (ms_m * 1e-3), "unit": "projections/s", "ms_per_step": ms_m,
                    "mapper": f"nn.Linear({cs.n}, {cs.n}) fused into the projection kernel",
                    "kernel": ("mfma_pair_wl_f16x2 with the mapper's image next to W's in LDS (round 6)"
                               if _lib.load().rayen_last_forward_kernel() == getattr(_lib, "KERNEL_PAIR_WL", -1)
                               else "mapped instances of the plain schedule"),
                    "how": "ConstraintModule(cs, input_dim=n, create_map=True) in eval mode, same batch, same step count; "
                           "`python bench.py --mapper D` times it as the main line"}
            except Exception as e:   # (a leg beside the measurement: never the reason the line is missing)
                out["module_with_mapper"] = {"error": repr(e)}
        if args.mapper:
            fused = on_gpu and (not args.no_fuse) and dtype == torch.float32 and dp.mapper_fusable(args.mapper)
            out["config"]["mapper"] = f"nn.Linear({args.mapper}, {cs.n}) " + ("fused into the projection kernel" if fused else "as its own GEMM")
        if split and world == 1 and not args.mapper and not args.no_families:
            # the same workload on the other fp32 families (RayenPackDesc.fp32_mode / RAYEN_FP32_MODE at pack
            # creation), timed the same way, so that one line carries all of them
            if served_by == getattr(_lib, "KERNEL_PAIR_WL", -1):
                # the same pack on the default schedule of rounds 3-5 (rows trickled through LDS, W streamed from L2): same values
                prev = _lib.load().rayen_pair_schedule(1)
                try:
                    with torch.no_grad():
                        for _ in range(SETTLE_LAUNCHES // 3):
                            module_step(x)
                    torch.cuda.synchronize()
                    _, ms = timed_loop(module_step, x, args.steps, args.warmup, False, graph=graph)
                finally:
                    _lib.load().rayen_pair_schedule(prev)
                tf = flops_pp * B / (ms * 1e-3) / 1e12
                out["pair_kernel_with_trickled_rows"] = {
                    "value": B / (ms * 1e-3), "unit": "projections/s", "ms_per_step": ms,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf},
                    "how": "rayen_pair_schedule(1): rayen_mfma_pair_io.hip (the default of rounds 3-5), same pack, same inputs, same step count"}
            if served_by == _lib.KERNEL_PAIR_IO:
                # the same pack on the plain f16-pair kernel (rows loaded / stored at the group boundaries): same values
                prev = _lib.load().rayen_pair_schedule(0)
                try:
                    with torch.no_grad():
                        for _ in range(SETTLE_LAUNCHES // 3):
                            module_step(x)
                    torch.cuda.synchronize()
                    _, ms = timed_loop(module_step, x, args.steps, args.warmup, False, graph=graph)
                finally:
                    _lib.load().rayen_pair_schedule(prev)
                tf = flops_pp * B / (ms * 1e-3) / 1e12
                out["pair_kernel_without_trickled_rows"] = {
                    "value": B / (ms * 1e-3), "unit": "projections/s", "ms_per_step": ms,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf},
                    "how": "rayen_pair_schedule(0): rayen_mfma_pair.hip, same pack, same inputs, same step count"}
            others = [("exact_fp32_kernels", "1", PEAK_FP32_TFLOPS, "exact-fp32 MFMA kernels (fp32_mode 1)")]
            if info.mfma_f32 == 3:
                others.insert(0, ("bf16_triple_kernel", "4", PEAK_BF16_TFLOPS / 6.0,
                                  "bf16-triple kernel, six piece products per fp32 product (fp32_mode 4)"))
            for key, mode, peak, how in others:
                os.environ["RAYEN_FP32_MODE"] = mode
                try:
                    other = ConstraintModule(cs, method="RAYEN", create_map=False).to(device)
                    other.check_nan = False
                    other.device_pack(device)
                finally:
                    del os.environ["RAYEN_FP32_MODE"]
                with torch.no_grad():
                    for _ in range(SETTLE_LAUNCHES // 3):    # first launches of this family's kernels; not part of W or K
                        other(x)
                torch.cuda.synchronize()
                _, ms = timed_loop(lambda xx: other(xx), x, args.steps, args.warmup, False, graph=graph)
                tf = flops_pp * B / (ms * 1e-3) / 1e12
                out[key] = {"value": B / (ms * 1e-3), "unit": "projections/s", "ms_per_step": ms,
                            "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak},
                            "how": how + ", same inputs, same step count"}
            out["fp32_family_check"] = {"pair_vs_fp64": info.fp32_check_pair, "triple_vs_fp64": info.fp32_check_split,
                                        "exact_vs_fp64": info.fp32_check_exact,
                                        "what": "worst row error on the pack-creation probe directions (-1: not measured)"}
        if on_gpu and world == 1 and not args.no_cpu_baseline and not args.mapper:
            out["cpu_baseline"] = cpu_baseline(raw, cs, B, dtype, args.cpu_seconds, rng)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        line = json.dumps(out)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (block-buffered when stdout is a file): flush it first so
        # that the JSON line is the LAST line of the output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        os.write(json_fd, (line + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
