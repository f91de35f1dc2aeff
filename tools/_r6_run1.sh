mkdir -p gpurun_out/r6a
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
export MELD_COMMIT=r6-wip
( python tools/make_fullsize_golden.py gpurun_out/r6a/g8_fullsize.npz 500000,1000000 > gpurun_out/r6a/golden.log 2>&1 ) 
echo "== vk"; bash tools/_vk.sh head 2>&1 | grep -v amdgpu.ids
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_partial_search.py tests/test_gpu_parity.py -x -q -k "frame or partial or 50k_config or odd_number or tile_pruning or direct_step" 2>&1 | tail -5
tail -3 gpurun_out/r6a/golden.log
