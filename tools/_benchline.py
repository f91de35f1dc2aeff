import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), {k: d["roofline"][k] for k in ("achieved","frac") if k in d["roofline"]}, "cheby", {k: round(d["roofline_cheby"][k], 4) for k in ("frac",) if "roofline_cheby" in d})
st = d.get("stages")
if st: print({k: round(1e3*v,2) for k,v in st.items()})
