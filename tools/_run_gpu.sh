python tools/cheby_only.py 1000000 2>&1 | grep -v amdgpu.ids
MELD_SPMM_RB=64 python tools/cheby_only.py 1000000 2>&1 | grep -v amdgpu.ids
