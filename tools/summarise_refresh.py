"""Writes gpurun_out/refresh/summary.txt from the files tools/refresh_profiles.sh left there: one
screen of the figures the kept profiles hold (bench line, prefill kernels, CPU legs, other model
shapes, per-rank compute, HBM traffic, rocprof kernel averages, micro-benchmarks)."""
import csv
import json
import os
import sys

O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/refresh"


def last_json(name):
    try:
        with open(os.path.join(O, name)) as f:
            return json.loads(f.read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None


out = []
d = last_json("bench.json")
if d:
    r = d["roofline"]
    out.append(f"8b: {d['value']:.1f} tok/s {d['ms_per_step']:.4f} ms  gemv {r['avg_launch_us']:.2f} us {r['achieved']:.0f} GB/s "
               f"frac {r['frac']:.4f}  traffic {r.get('traffic')} ({r.get('traffic_source')})  e2e "
               f"{d.get('decode_hbm_gbps_end_to_end', 0):.0f} GB/s")
    out.append(f"  prefill {d.get('prefill_tok_per_s') or 0:.0f} tok/s with kept fp16 images "
               f"({(d.get('prefill_weight_image_bytes_per_rank') or 0) / 1e9:.2f} GB), "
               f"{d.get('prefill_int8_only_tok_per_s') or 0:.0f} tok/s on the int8 weights alone")
    for k, v in (d.get("prefill_kernels") or {}).items():
        if isinstance(v, dict):
            out.append(f"  {k}: {v['us']:.1f} us {v['tflops']:.0f} TF {v['mfma_util']:.3f}")
    c = d.get("cpu_baseline")
    if c:
        out.append(f"  cpu {c['value']:.2f} tok/s {c['cores']} cores {c.get('gemv_gbps_all_cores', 0):.1f} GB/s, "
                   f"1 thread {c.get('tok_per_s_1_thread', 0):.3f}")
    c1 = d.get("config1")
    if c1:
        out.append(f"  config1 (stories110M fp32 CPU token loop): {c1['steps_per_s_all_cores']:.1f} steps/s on "
                   f"{c1['cores']} cores, {c1['steps_per_s_1_thread']:.1f} on 1 thread")
for name in ("b70", "b1", "bq", "bs", "bkv16", "b8k", "bkv16_8k"):
    d = last_json(name + ".json")
    if d:
        r = d["roofline"]
        out.append(f"{name}: {d['value']:.1f} tok/s {d['ms_per_step']:.3f} ms  gemv {r['avg_launch_us']:.2f} us "
                   f"{r['achieved']:.0f} GB/s {r['frac']:.3f}  traffic {r.get('traffic')}")
d = last_json("tp.json")
if d:
    for k, v in d["per_rank"].items():
        out.append(f"{k}: gemv {v['gemv_us_per_launch']:.2f} us attn {v['attn_us']:.2f} us token {v['compute_us_per_token']:.0f} us")
d = last_json("traffic.json")
if d:
    out.append(f"traffic {d.get('hbm_bytes_per_launch'):.0f} B/launch")
for sub, name in (("prof", "b_kernel_stats.csv"), ("prof_prefill", "p_kernel_stats.csv")):
    try:
        rows = list(csv.DictReader(open(os.path.join(O, sub, name))))
    except OSError:
        continue
    gemv_ns = gemv_calls = 0
    for row in rows[:8]:
        out.append(f"  {row['Name'][:90]}: {row['Calls']} x {float(row['AverageNs']) / 1e3:.2f} us")
    for row in rows:
        if "gemv_w8_kernel" in row["Name"] or "gemv_w8_pick_kernel" in row["Name"]:
            gemv_ns += float(row["TotalDurationNs"])
            gemv_calls += int(row["Calls"])
    if gemv_calls:
        out.append(f"rocprof gemv avg {gemv_ns / gemv_calls / 1e3:.2f} us over {gemv_calls} launches")
for name in ("microbench.txt", "short_prompts.txt", "few_rows_8b.txt", "prefill_model.txt"):
    try:
        out.append(open(os.path.join(O, name)).read().rstrip())
    except OSError:
        pass
with open(os.path.join(O, "summary.txt"), "w") as f:
    f.write("\n".join(out) + "\n")
print("\n".join(out[:24]))
