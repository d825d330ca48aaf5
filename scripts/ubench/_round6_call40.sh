out=gpurun_out/r06zv; mkdir -p $out
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^$" | head -30 > $out/smi_before.txt
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-families > $out/bench_c3_$i.json 2>/dev/null; python -c "
import json;d=json.loads(open('$out/bench_c3_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['kernel_ms'], d['training_step']['backward_ms'], d['training_step']['forward_with_record_ms'])"; done
for i in 1 2; do timeout 200 python scripts/ubench/wl_check.py 2>&1 | grep -v amdgpu | tail -3; done
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^$" | head -30 > $out/smi_after.txt
cat $out/smi_after.txt
