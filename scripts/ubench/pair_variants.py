"""Build librayen_hip variants that differ in RAYEN_PAIR_VARIANT (the developer copy of the f16-pair kernel with its
experiment switches, scripts/ubench/experiments/rayen_mfma_pair_variants.hip.txt: 1 streaming y stores everywhere,
2 / 128 L2 warm-up of the next group's rows, 4 streaming v loads, 8 no A stream (timing only), 16 / 32 asymmetric
burst priority, 64 phase timestamps over kappa_out, 64 + 256 = 320 per-tile timestamps (tile top / burst issued /
epilogue done, from tile 5 on); read by scripts/ubench/pair_stamps.py) into
rayen_amd/csrc/variants/ and, on a GPU, time config 3 / 5 with each.
    python scripts/ubench/pair_variants.py build 1 2 3 ...     (here)
    python scripts/ubench/pair_variants.py run 0 1 2 3 ...       (GPU box; 0 = the product library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "rayen_amd", "csrc")
VAR = os.path.join(CSRC, "variants")
mode, ids = sys.argv[1], sys.argv[2:]
if mode == "build":
    sys.path.insert(0, ROOT)
    from rayen_amd import _build
    _build.build()
    os.makedirs(VAR, exist_ok=True)
    objs = [os.path.join(CSRC, "_obj", os.path.splitext(s)[0] + ".o") for s in _build.SOURCES if s != "rayen_mfma_pair.hip"]
    import shutil
    src = os.path.join(VAR, "pair_variant_src.hip")
    shutil.copy(os.path.join(ROOT, "scripts", "ubench", "experiments", "rayen_mfma_pair_variants.hip.txt"), src)
    for i in ids:
        obj = os.path.join(VAR, f"pair_v{i}.o")
        subprocess.run([_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                        "-I", CSRC, f"-DRAYEN_PAIR_VARIANT={i}", "-c", src, "-o", obj], check=True)
        subprocess.run([_build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, obj, "-o",
                        os.path.join(VAR, f"librayen_hip_v{i}.so")], check=True)
        print("built variant", i)
else:
    for cfg in ("c3", "c5"):
        for i in ids:
            env = dict(os.environ)
            if i != "0":
                env["RAYEN_HIP_LIBRARY"] = os.path.join(VAR, f"librayen_hip_v{i}.so")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--no-cpu-baseline", "--no-families"],
                                 env=env, capture_output=True, text=True)
            try:
                line = json.loads(out.stdout.strip().splitlines()[-1])
                print(cfg, "variant", i, "ms", round(line["ms_per_step"], 5), "kernel_ms", round(line["roofline"]["kernel_ms"], 5),
                      line["config"]["kernel"], "viol", line["max_violation"], flush=True)
            except Exception as exc:
                print(cfg, "variant", i, "failed", exc, out.stderr[-400:], flush=True)
