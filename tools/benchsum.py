#!/usr/bin/env python3
"""Compact view of a bench.py JSON line: headline, per-launch table, text numbers.   python tools/benchsum.py FILE"""
import json, sys
d = json.load(open(sys.argv[1]))
print(f'steps/s {d["value"]}  ms/step {d["ms_per_step"]}  steady {d.get("steady_state")}')
lt = d.get("launch_table") or {}
if lt:
    print("sum_us", lt["sum_us"], "conv_blocks", lt["conv_blocks"])
    print(" ".join(f'{k}:{v}' for k, v in lt["all_us"].items()))
r = d.get("roofline") or {}
print("roofline", {k: r.get(k) for k in ("frac", "achieved", "ms_per_launch", "traffic")}, (r.get("kernel") or "")[:80])
s = d.get("sampling") or {}
print("sampling", s.get("ms_per_reverse_step"), s.get("imgs_per_s_1000_step"), s.get("frac_hbm"))
t = d.get("text_denoiser") or {}
print("text", t.get("ms_per_step"), "bf16:", (t.get("other_gemm_mode") or {}).get("ms_per_step"), "head:", (t.get("rounding_head") or {}).get("ms"),
      ((t.get("rounding_head") or {}).get("stored_logits_form") or {}).get("ms"))
