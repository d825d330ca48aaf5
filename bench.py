#!/usr/bin/env python
"""Throughput of the RAYEN projection on MI355X -- the driver's bench contract.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (``ConstraintModule.forward`` with the identity
mapper = one launch of the fused projection) over one batch of synthetic directions
that is already resident in HBM.  Default workload: BASELINE.json ``configs[2]``, the
headline -- k=64, 128 linear + 4 quadratic + 2 SOC constraints, batch 262144 per GPU,
fp32.  N>1 runs one process per GPU (torchrun, backend nccl = RCCL); the batch
dimension is sharded with fixed per-GPU work (weak scaling) and no data-path
collective (samples are independent; ``--gather`` adds the all-gather of ``y`` the
north_star describes for a caller that wants every output on every rank).

One JSON line on rank 0: metric / value (whole-job projections/s) plus
``roofline`` (dominant kernel, live HIP-event timing) and ``cpu_baseline`` (the
PyTorch-CPU oracle, same workload, timed on this box's host cores).
"""
from __future__ import annotations

import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = the BASELINE.json size)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--gather", action="store_true", help="all-gather y across ranks inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank (exercises the multi-GPU code path)")
    ap.add_argument("--mapper", type=int, default=0, metavar="D",
                    help="put the module's nn.Linear(D, n) mapper in front (create_map=True); 0 = identity mapper")
    ap.add_argument("--no-fuse", action="store_true", help="with --mapper: run the mapper as its own GEMM")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(raw, cs, B, dtype, budget_s):
    """Reference op sequence (oracle/rayen_oracle.py) on the host cores, bounded sample of the workload."""
    from oracle import rayen_oracle as oracle
    cores = os.cpu_count() or 1
    csd = {key: getattr(cs, key) for key in ("A_p", "b_p", "NA_E", "yp", "z0", "y0")}
    csd.update(P=raw["P"], q=raw["q"], r=raw["r"], M=raw["M"], s=raw["s"], c=raw["c"], d=raw["d"],
               F=raw["F"])
    buf = oracle.precompute(csd, dtype)
    gen = torch.Generator().manual_seed(1234)
    Bs = min(B, 32768)
    x = torch.empty(Bs, cs.n, 1, dtype=dtype).uniform_(-1.0, 1.0, generator=gen)

    def timed(xx):
        t0 = time.perf_counter()
        oracle.forward(buf, xx)
        return time.perf_counter() - t0

    with torch.no_grad():
        # PyTorch-CPU does not scale to every core on this op mix: probe a few thread counts, keep the best
        probe = x[:4096]
        rates = {}
        for threads in sorted({t for t in (4, 8, 16, 32, 64, cores) if t <= cores}):
            torch.set_num_threads(threads)
            timed(probe)
            rates[threads] = probe.shape[0] / min(timed(probe), timed(probe))
        threads = max(rates, key=rates.get)
        torch.set_num_threads(threads)
        timed(x)
        best, reps, t_start = float("inf"), 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t_start < budget_s and reps < 30):
            best = min(best, timed(x))
            reps += 1
    return {"value": Bs / best, "unit": "projections/s", "cores": threads, "kind": "port",
            "host_cores": cores,
            "sample": f"B={Bs} slice of the same workload, best of {reps} calls after warm-up, "
                      f"{str(dtype).split('.')[-1]}, torch {torch.__version__} CPU, {threads} threads "
                      f"(best of thread counts {sorted(rates)})"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from rayen_amd import workloads
    from rayen_amd.constraint_module import ConstraintModule
    from rayen_amd import ops

    dtype = torch.float32 if args.dtype == "fp32" else torch.float64
    torch.set_default_dtype(dtype)
    raw = workloads.make_raw(args.config, seed=0)
    cs = workloads.build_constraints(raw)
    if args.mapper:
        torch.manual_seed(0)
        layer = ConstraintModule(cs, input_dim=args.mapper, method="RAYEN", create_map=True).to(device)
        layer.fuse_mapper = not args.no_fuse
    else:
        layer = ConstraintModule(cs, method="RAYEN", create_map=False).to(device)
    layer.check_nan = False                      # no host sync inside the timed region
    B = args.batch or workloads.CONFIGS[args.config][2]
    if args.config == "c5" and not args.batch:
        B = B // 8                               # 2M over 8 GPUs -> 262144 per GPU
    rng = workloads.CONFIGS[args.config][3]
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    x = torch.empty(B, args.mapper or cs.n, 1, device=device, dtype=dtype).uniform_(-rng, rng, generator=gen)
    gathered = torch.empty(world * B, cs.k, 1, device=device, dtype=dtype) if (args.gather and use_dist) else None

    def step():
        y = layer(x)
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, y)
        return y

    with torch.no_grad():
        for _ in range(args.warmup):
            y = step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            y = step()
        ev1.record()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps    # HIP events on the launch stream

    t = torch.tensor([elapsed, dev_ms], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, dev_ms = float(t[0]), float(t[1])

    # feasibility of what was just computed (fp64 residuals on a slice, outside the timed region)
    from rayen_amd.constraints import ConvexConstraints  # noqa: F401
    sl = y[: min(B, 65536), :, 0].double().cpu().numpy()
    max_violation = cs.getMaxViolation(sl)

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
        info = layer.device_pack(device)[0].info()
        split = dtype == torch.float32 and info.mfma_f32 == 2
        # The split-operand kernel rebuilds every fp32 product from six bf16 MFMA products (fp32-grade results),
        # so its matrix ceiling in ALGORITHMIC fp32 flops is the dense bf16 peak / 6, not the fp32 MFMA peak
        peak_tf = (PEAK_BF16_TFLOPS / 6.0 if split else PEAK_FP32_TFLOPS) if dtype == torch.float32 else PEAK_FP64_TFLOPS
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        if ai > ridge:
            roof = {"bound": "mfma", "achieved": tflops, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": tflops / peak_tf, "traffic": None}
            if split:
                roof["peak_basis"] = "dense bf16 MFMA peak %.1f / 6 piece products per fp32 product" % PEAK_BF16_TFLOPS
                roof["frac_of_fp32_mfma_peak"] = tflops / PEAK_FP32_TFLOPS
        else:
            roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": gbs / PEAK_HBM_GBS, "traffic": None}
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE x2 per the
        # gfx950 correction + WRITE_SIZE), committed under profiles/ by scripts/summarize_profile.py;
        # bench.py itself cannot collect PMC counters, so this is null for workloads never profiled
        if args.config == "c3" and dtype == torch.float32 and B == 262144:
            import glob
            found = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_c3_split_final_rocprofv3.json" if split
                                                  else "r*_c3_mfma_final_rocprofv3.json")))
            if found:
                prof = json.load(open(found[-1]))
                roof["traffic"] = prof.get("derived", {}).get("hbm_bytes_per_launch")
                roof["traffic_unit"] = "bytes/launch (algorithmic: %d)" % (bytes_pp * B)
                roof["traffic_source"] = os.path.relpath(found[-1], REPO)
        roof.update({"kernel_ms": dev_ms, "algorithmic_flops_per_projection": flops_pp,
                     "algorithmic_bytes_per_projection": bytes_pp, "hbm_GBps": gbs,
                     "hbm_frac": gbs / PEAK_HBM_GBS, "TFLOPs": tflops})
        out = {
            "metric": "feasible projections/sec at k=64, 128 lin+4 quad+2 SOC; max violation"
                      if args.config == "c3" else f"feasible projections/sec ({args.config})",
            "value": world * B * args.steps / elapsed,
            "unit": "projections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: k={cs.k} n={cs.n}, {cs.A_p.shape[0]} linear + "
                                   f"{len(cs.qcs)} quadratic + {len(cs.socs)} SOC"
                                   f"{' + 1 LMI' if cs.has_lmi_constraints else ''}, "
                                   f"batch {B} per GPU, v~U(-{rng:g},{rng:g})",
                       "batch_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"batch-sharded x{world}" + (" + all-gather(y)" if gathered is not None else ""),
                       "kernel": ({2: "mfma_split_bf16x3 (fp32-grade)", 1: "mfma_f32"}.get(info.mfma_f32, "generic")) if dtype == torch.float32
                       else ("mfma_f64" if info.mfma_f64 else "generic")},
            "max_violation": max_violation,
            "violations_gt_1e-6": int(max_violation > 1e-6),
            "roofline": roof,
        }
        if args.mapper:
            fused = (not args.no_fuse) and dtype == torch.float32 and layer.device_pack(device)[0].mapper_fusable(args.mapper)
            out["config"]["mapper"] = f"nn.Linear({args.mapper}, {cs.n}) " + ("fused into the projection kernel" if fused else "as its own GEMM")
        if split and world == 1 and not args.mapper:
            # the same workload on the exact-fp32 MFMA kernels (RAYEN_SPLIT_BF16=0 is read when a pack is created),
            # timed the same way, so that one line carries both fp32 families
            os.environ["RAYEN_SPLIT_BF16"] = "0"
            try:
                exact = ConstraintModule(cs, method="RAYEN", create_map=False).to(device)
                exact.check_nan = False
                with torch.no_grad():
                    for _ in range(args.warmup):
                        exact(x)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.steps):
                        exact(x)
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.steps
                tf = flops_pp * B / (ms * 1e-3) / 1e12
                out["exact_fp32_kernels"] = {"value": B / (ms * 1e-3), "unit": "projections/s", "ms_per_step": ms,
                                             "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS,
                                                          "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS},
                                             "how": "RAYEN_SPLIT_BF16=0, same inputs, same step count"}
            finally:
                del os.environ["RAYEN_SPLIT_BF16"]
        if world == 1 and not args.no_cpu_baseline and not args.mapper:
            out["cpu_baseline"] = cpu_baseline(raw, cs, B, dtype, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
