#!/usr/bin/env python3
"""profiles/traffic.json from a summary written by tools/gpu_profile.sh: FETCH_SIZE / WRITE_SIZE (KB per dispatch, their
own --pmc passes) of the timed render kernel.  usage: traffic_json.py <summary.txt> <name the summary is committed under>"""
import json
import re
import sys

text = open(sys.argv[1]).read()
vals = {}
for line in text.splitlines():
    # the timed frames go through one instantiation that does not count work (first template argument false)
    m = re.match(r"(k_render_(?:jobs|direct)<false, [^>]*>)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", line)
    if m and int(m.group(3)) > vals.get(m.group(2), (0, 0, ""))[1]:
        vals[m.group(2)] = (float(m.group(4)), int(m.group(3)), m.group(1))
bench = [json.loads(ln) for ln in text.splitlines() if ln.startswith('{"metric"')]
wl = bench[-1]["config"]["workload"].split(" recursion_limit")[0] if bench else "?"
print(json.dumps({
    "workload": wl, "parts": 1, "kernel": vals.get("FETCH_SIZE", (0, 0, "?"))[2],
    "fetch_kb": vals.get("FETCH_SIZE", (None,))[0], "write_kb": vals.get("WRITE_SIZE", (None,))[0],
    "source": f"{sys.argv[2]} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, per dispatch average of "
              f"{vals.get('FETCH_SIZE', (0, 0))[1]})",
    "note": "bench.py doubles fetch_kb (gfx950 FETCH_SIZE counts 64 B per 128-B request) and adds write_kb"}, indent=1))
