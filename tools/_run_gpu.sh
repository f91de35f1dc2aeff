python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | cut -c1-200
cp meld_amd/libmeld_hip.so /tmp/orig.so
for w in 8 12; do
cp meld_amd/libmeld_hip_w$w.so meld_amd/libmeld_hip.so
echo "--- $w-wave workgroups"
python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
