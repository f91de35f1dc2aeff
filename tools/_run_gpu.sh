mkdir -p gpurun_out/r2r
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2r/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2r/pytest.log
tail -12 gpurun_out/r2r/pytest.log
{
python tools/time_pca.py 200000 2000 50
python tools/time_pca.py 1000000 200 50
python tools/time_pca.py 100000 12000 100
python bench.py --steps 5 --warmup 1 --cpu-sample 0 --stages
} > gpurun_out/r2r/timing.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2r/timing.log | cut -c1-330; grep -o '"stages.*' gpurun_out/r2r/timing.log | cut -c1-600
