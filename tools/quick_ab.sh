python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_robustness.py -x -q -m gpu > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log; python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming --no-images > gpurun_out/b1.json 2> gpurun_out/b1.err; python - <<EOP
import json
d=json.loads(open("gpurun_out/b1.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["end_to_end"]["fps"])
EOP
