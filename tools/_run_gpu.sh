mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2m/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2m/pytest.log
tail -5 gpurun_out/r2m/pytest.log
python bench.py --steps 5 --warmup 1 --cpu-sample 0 --stages > gpurun_out/r2m/bench.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2m/bench.log | cut -c1-400; grep -o '"stages.*' gpurun_out/r2m/bench.log
python tools/profile_transform.py > gpurun_out/r2m/transform.log 2>&1; tail -22 gpurun_out/r2m/transform.log | cut -c1-150
