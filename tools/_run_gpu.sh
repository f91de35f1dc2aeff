python tools/time_vfc.py 600 2>&1 | grep -v amdgpu.ids | tail -1
python tools/time_vfc.py 4000 2>&1 | grep -v amdgpu.ids | tail -1
python tools/time_vfc.py 12000 2>&1 | grep -v amdgpu.ids | tail -1
