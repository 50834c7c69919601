#!/usr/bin/env python3
"""one-line summary of a bench.py JSON line: tools/print_bench.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("images/s", d["value"], "ms/step", d["ms_per_step"], "one-in-flight", (d.get("one_batch_in_flight") or {}).get("value"),
      "roofline", (d.get("roofline") or {}).get("frac"), {k: v["value"] for k, v in (d.get("other_configs") or {}).items()})
