"""Compact view of bench.py's JSON line (stdin): value, ms/step, kernel ms, roofline fraction."""
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print("value %.0f %s  ms/step %.4f  kernel_ms %.4f  frac %.3f  (%s)" % (
        d["value"], d["unit"], d["ms_per_step"], r.get("kernel_ms", 0), r.get("frac", 0), d["config"].get("path", "")))
