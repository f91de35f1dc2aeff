export MELD_DEV=1
python bench.py --cells 1000000 --dims 100 --steps 3 --no-extra --cpu-sample 0 --no-host-input --stages 2>/dev/null | python tools/_benchline.py
python bench.py --cells 1000000 --dims 64 --steps 3 --no-extra --cpu-sample 0 --no-host-input 2>/dev/null | python tools/_benchline.py
python bench.py --cells 1000000 --dims 80 --steps 3 --no-extra --cpu-sample 0 --no-host-input 2>/dev/null | python tools/_benchline.py
timeout 600 python -m pytest tests/test_gpu_api.py -x -q 2>&1 | tail -2
