mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2z/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2z/pytest.log
tail -4 gpurun_out/r2z/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py 2>gpurun_out/r2z/bench.err | cut -c1-400
