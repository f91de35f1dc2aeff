mkdir -p gpurun_out/r6b
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --cpu-sample 0 --no-host-input --stages 2>gpurun_out/r6b/bench.err | tee gpurun_out/r6b/bench.json | python tools/_benchline.py
