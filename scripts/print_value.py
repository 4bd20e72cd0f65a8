import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d.get("forwards_in_flight"), round(d["value"], 1), round(d["ms_per_step"], 2), d["sanity"]["median_rot_err_vs_gt"])
