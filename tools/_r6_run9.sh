export MELD_DEV=1
for r in 0 1 0 1; do echo "== regstage $r"; MELD_KNN_FILTER_REGSTAGE=$r python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | grep "knn_filter" | tail -1; done
timeout 300 python -m pytest tests/test_gpu_partial_search.py -x -q 2>&1 | tail -1
MELD_KNN_FILTER_REGSTAGE=1 timeout 300 python -m pytest tests/test_gpu_partial_search.py -x -q 2>&1 | tail -1
python tools/save_graph.py 1000000 /tmp/g1m.pt > /dev/null 2>&1; python tools/spmm_time.py /tmp/g1m.pt 2>/dev/null | grep "tiled p\|plain copy"
