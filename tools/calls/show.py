"""print the essentials of a bench.py JSON line: python tools/calls/show.py <file> [label]"""
import json
import sys

try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:      # noqa: BLE001
    print(sys.argv[1], "unreadable:", e)
    sys.exit(0)
r = j.get("roofline") or {}
print(sys.argv[2] if len(sys.argv) > 2 else "", "ms/step", round(j["ms_per_step"], 4), "| alone us:", r.get("kernels_us_per_step"))
tl = r.get("in_step_timeline_us")
if tl:
    print("   in-step:", "  ".join(f"{k}@{v['start_us']}+{v['dur_us']}" for k, v in sorted(tl.items(), key=lambda kv: kv[1]["start_us"])))
if "also" in j:
    a = j["also"].get("configs[4]", {})
    print("   also configs[4]:", a.get("ms_per_step"), a.get("error"))
    b = j["also"].get("shipped_conv_gp") or {}
    print("   also shipped conv+gp:", b.get("ms_per_step"), b.get("error"))
