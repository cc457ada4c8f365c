"""Extract the reference's two FIXED landmark maps (data, not code) into live_ekf_slam_amd/data/fixed_maps.json.

Runs only in the build container (needs /root/reference): `demo_map` (sim_node.py:26-30, map_type "demo") and the
`barrels` list of map_type "igvc1" (sim_node.py:190).  The values are parsed from the source text with `ast.literal_eval`;
nothing of the reference is imported or copied besides these coordinates."""
import ast, json, os, re

SRC = "/root/reference/ekf_ws/src/base_pkg/src/sim_node.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "live_ekf_slam_amd", "data", "fixed_maps.json")
txt = open(SRC).read()
demo = ast.literal_eval(re.search(r"demo_map\s*=\s*(\{.*?\})", txt, re.S).group(1))
barrels = ast.literal_eval(re.search(r"barrels\s*=\s*(\[.*?\])\n", txt, re.S).group(1))
out = {"source": "kevin-robb/live_ekf_slam ekf_ws/src/base_pkg/src/sim_node.py:26-30 (demo_map), :190 (igvc1 barrels); id = list index",
       "demo": [list(demo[i]) for i in sorted(demo)], "igvc1": [list(p) for p in barrels]}
json.dump(out, open(OUT, "w"), indent=1)
print(len(out["demo"]), len(out["igvc1"]))
