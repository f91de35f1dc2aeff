mkdir -p gpurun_out/r2v
{
python tools/knn_only.py 1000000 3
MELD_KNN16_TWO_SIDED=0 python tools/knn_only.py 1000000 3
MELD_KNN16_BATCH_EVERY=32 python tools/knn_only.py 1000000 3
MELD_KNN16_ABLATION=1 python tools/knn_only.py 1000000 2
MELD_KNN16_ABLATION=3 python tools/knn_only.py 1000000 2
MELD_KNN16_ABLATION=6 python tools/knn_only.py 1000000 2
} > gpurun_out/r2v/timing.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2v/timing.log | cut -c1-330
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2v/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2v/pytest.log
tail -4 gpurun_out/r2v/pytest.log
