export MELD_DEV=1
python tools/_probe_d100.py 100 2>&1 | grep -v amdgpu.ids | tail -12
