out=gpurun_out/r06zzg; mkdir -p $out
for i in 1 2 3; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c3_$i.json 2> $out/bench_c3_$i.err
python -c "
import json;d=json.loads(open('$out/bench_c3_$i.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['hbm_GBps'], d['roofline']['l3_resident']['kernel_ms'], d['training_step']['forward_with_record_ms'], d['training_step']['backward_ms'], d['pair_kernel_with_trickled_rows']['ms_per_step'])"
done
