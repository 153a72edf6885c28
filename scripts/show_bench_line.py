#!/usr/bin/env python3
"""Prints the headline fields of a bench.py JSON line read from stdin."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f %s  ms/step %.3f" % (d["value"], d["unit"], d["ms_per_step"]))
for k in ("host_entry", "cpu_baseline", "roofline"):
    if k in d:
        print(k, json.dumps(d[k]))
