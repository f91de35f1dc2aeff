export MELD_DEV=1
timeout 1500 python -m pytest tests/test_gpu_partial_search.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "frame or partial or oracle_digest" 2>&1 | tail -3
mkdir -p gpurun_out/r6c; timeout 1200 python tools/fuzz_partial.py 60 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r6c/fuzz_partial.txt; tail -4 gpurun_out/r6c/fuzz_partial.txt | cut -c1-220; grep -c "^ok" gpurun_out/r6c/fuzz_partial.txt; grep -c "^BAD" gpurun_out/r6c/fuzz_partial.txt
