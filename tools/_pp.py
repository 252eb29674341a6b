import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d["steps"]
print("ms_per_step %.1f" % d["ms_per_step"], {k: round(v / st, 1) for k, v in d["kernels_ms"].items() if k.startswith("sw_fwd")}, "c4", (d.get("config4_profile_targets") or {}).get("s_per_pass"), (d.get("config4_profile_targets") or {}).get("result_digest", {}).get("match"))
