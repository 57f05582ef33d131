#!/usr/bin/env python3
"""Prints the figures of gpurun_out/refresh/ (tools/refresh_profiles.sh) in one screen."""
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/refresh"


def load(name):
    with open(os.path.join(d, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


m = load("bench.json")
r = m["roofline"]
print(f"8b: {m['value']:.1f} tok/s {m['ms_per_step']:.4f} ms  gemv {r['avg_launch_us']:.2f} us "
      f"{r['achieved']:.0f} GB/s frac {r['frac']:.4f}  e2e {m['decode_hbm_gbps_end_to_end']:.0f} GB/s  "
      f"prefill {m['prefill_tok_per_s']:.0f}")
for k, v in m["prefill_kernels"].items():
    if isinstance(v, dict):
        print(f"  {k}: {v['us']:.1f} us {v['tflops']:.0f} TF {v['mfma_util']:.3f}")
c = m["cpu_baseline"]
print(f"  cpu {c['value']:.2f} tok/s {c['cores']} cores {c['gemv_gbps_all_cores']:.1f} GB/s, 1 thread {c['tok_per_s_1_thread']:.3f}")
for f in ("b70", "b1", "bq", "bs", "bkv16", "b8k", "bkv16_8k"):
    x = load(f + ".json")
    r = x["roofline"]
    print(f"{f}: {x['value']:.1f} tok/s {x['ms_per_step']:.3f} ms  gemv {r['avg_launch_us']:.2f} us {r['achieved']:.0f} GB/s {r['frac']:.3f}")
for k, v in load("tp.json")["per_rank"].items():
    print(f"{k}: gemv {v['gemv_us_per_launch']:.2f} us attn {v['attn_us']:.2f} us token {v['compute_us_per_token']:.0f} us")
t = json.load(open(os.path.join(d, "traffic.json")))
print(f"traffic {t['hbm_bytes_per_launch']:.0f} B/launch")
tot = n = 0
with open(os.path.join(d, "prof", "b_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
        if "gemv_w8_kernel" in row["Name"]:
            tot += int(row["TotalDurationNs"]); n += int(row["Calls"])
        if "attn_decode" in row["Name"] or "attn_prefill" in row["Name"] or "pipe" in row["Name"]:
            print(f"  {row['Name'][:60]}: {row['Calls']} x {float(row['AverageNs'])/1e3:.2f} us")
print(f"rocprof gemv avg {tot / n / 1e3:.2f} us over {n} launches")
print(open(os.path.join(d, "microbench.txt")).read())
