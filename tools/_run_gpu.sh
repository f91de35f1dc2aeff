mkdir -p gpurun_out/r2t
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2t/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2t/pytest.log
tail -4 gpurun_out/r2t/pytest.log
python bench.py --steps 5 --warmup 1 --cpu-sample 0 --stages > gpurun_out/r2t/timing.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2t/timing.log | cut -c1-330; grep -o '"stages.*' gpurun_out/r2t/timing.log | cut -c1-600
