out=gpurun_out/r06zzb; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pair_wl.py tests/test_gpu_backward_dense_pairs.py tests/test_gpu_backward_pairs.py tests/test_gpu_backward.py tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_c3_$i.json 2> $out/bench_c3_$i.err
python -c "
import json;d=json.loads(open('$out/bench_c3_$i.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['l3_resident']['kernel_ms'], d['training_step']['forward_with_record_ms'], d['training_step']['backward_ms'], d['pair_kernel_with_trickled_rows']['ms_per_step'])"
done
