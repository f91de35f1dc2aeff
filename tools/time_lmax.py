"""Time the lmax estimate and the layout build on a saved graph: python tools/time_lmax.py graph.pt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch
from meld_amd.graph import DeviceGraph, HipOps

d = torch.load(sys.argv[1])
for rep in range(4):
    G = DeviceGraph(d["rowptr"].cuda(), d["col"].cuda(), d["val"].cuda(), d["dw"].cuda())
    G.ops = HipOps()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G.ops.pt_layout(G)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    G.estimate_lmax()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("layout %.2f ms  lmax %.2f ms (%d its)  %s" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), G.lmax_info["iterations"], {k: v for k, v in os.environ.items() if k.startswith("MELD_")}))
