out=gpurun_out/r06zzl; mkdir -p $out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err
python -c "
import json;d=json.loads(open('$out/bench_c3.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']); print(d['training_step']); print(d.get('module_with_mapper')); print(d['config']['kernel'])"
tail -3 $out/bench_c3.err
