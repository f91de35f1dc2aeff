mkdir -p gpurun_out/r2y
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2y/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2y/pytest.log
tail -4 gpurun_out/r2y/pytest.log
python bench.py --steps 3 --warmup 1 --cpu-sample 0 --force-sharded > gpurun_out/r2y/sharded.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2y/sharded.log | cut -c1-400
python bench.py --steps 5 --warmup 1 --cpu-sample 0 --stages > gpurun_out/r2y/bench.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2y/bench.log | cut -c1-300; grep -o '"stages.*' gpurun_out/r2y/bench.log | cut -c1-600
