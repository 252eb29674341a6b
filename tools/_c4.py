import json, sys
d = json.load(open(sys.argv[1]))
s = d["steps"][-1]
print({k: v for k, v in s.items() if k in ("prefilter_s", "align_s", "total_s")}, {k: v for k, v in s["kernels_ms"].items() if k.startswith("sw_fwd")})
