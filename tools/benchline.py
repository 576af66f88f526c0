"""Prints the few numbers of a bench.py JSON line that A/B runs look at:  python bench.py ... | python tools/benchline.py [label]"""
import json
import sys

label = " ".join(sys.argv[1:])
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        extra = ""
        if "roofline_large" in d:
            extra += f"  large {d['roofline_large']['kernel_us']:.1f} us {d['roofline_large']['frac']:.3f}"
        if "roofline_steady" in d:
            extra += f"  steady {d['roofline_steady']['kernel_us']:.2f} us {d['roofline_steady']['frac']:.3f}"
        print(f"{label}: {d['value'] / 1e9:.3f} G env-steps/s  {d['ms_per_step'] * 1e3:.3f} us/step  kernel {r['kernel_us']:.3f} us  "
              f"frac {r['frac']:.3f}  traffic {r.get('traffic')}{extra}")
