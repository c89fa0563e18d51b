#!/usr/bin/env python
"""Markdown table of DESIGN.md section 7 from profiles/r06_bench.json (tools/refresh_profiles_r06.py)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
rows = [("c4_fp16_n300_b8_driver_command", "**c4, the driver's command** (`--steps 20 --warmup 5`, de novo N = 300, 8 samples/GPU)"),
        ("c4_fp16_n300_b8_whole_trajectory", "c4, whole T = 500 trajectory (default `bench.py`)"), ("c4_fp16_eager", "... launch by launch (`--eager`)"),
        ("c4_fp16_b64", "c4, all 64 samples on one GPU (40 steps)"), ("c4_fp32", "c4 in the fp32 mode (reference arithmetic)"),
        ("c2_fp16_n128_b8", "c2: de novo N = 128, 8 samples"), ("c3_fp16_mixed_bucket_of_8_complexes", "c3: bucket of 8 complexes, 746 - 766 residues (padded)"),
        ("c3e_fp16_equal_length_n776_b8", "c3e: 8 samples of one 776-residue complex"), ("c3w_fp16_mixed_whole_range_700_850", "c3w: mixed lengths 700 - 850"),
        ("c5_fp32_n1000_b4", "c5: inpainting N = 1000, 4 samples"), ("c5_shape_in_fp16", "c5's shape in the fp16 mode"),
        ("c4_fp16_b24", "c4, 24 samples per GPU (100 steps)"), ("c4_fp16_without_split_operands", "c4 without split operands (`--kernel-flags 32`: outside the parity bar)")]
print("| config | mode | residue·step/s | ms/step | EdgeTransition (ms, frac, in-kernel GHz) | whole forward of the MFMA peak |")
print("|---|---|---|---|---|---|")
for k, name in rows:
    if k not in L:
        continue
    v, r = L[k], L[k]["roofline"]
    val = f"{v['value'] / 1e6:.3f} M" if v["value"] > 3e5 else f"{v['value'] / 1e3:.1f} k"
    ghz = f", {r['clock_ghz']:.2f}" if r.get("clock_ghz") else ""
    print(f"| {name} | {v['dtype']} | {val} | {v['ms_per_step']:.3f} | {r['avg_launch_ms']:.3f}, {r['frac']:.3f}{ghz} | {r['whole_forward_frac']:.3f} |")
    if k == "c4_fp16_n300_b8_driver_command":
        a, f = v["all_samples_one_gpu"], v["reference_precision"]
        print(f"| ... `all_samples_one_gpu` on the same JSON line (64 samples, 8 steps) | fp16 | {a['value'] / 1e6:.3f} M | {a['ms_per_step']:.2f} | {a['roofline']['avg_launch_ms']:.3f}, {a['roofline']['frac']:.3f} | {a['roofline']['whole_forward_frac']:.3f} |")
        print(f"| ... `reference_precision` on the same JSON line (6 steps) | fp32 | {f['value'] / 1e3:.1f} k | {f['ms_per_step']:.2f} | {f['roofline']['avg_launch_ms']:.3f}, {f['roofline']['frac']:.3f} of the fp32 peak | {f['roofline']['whole_forward_frac']:.3f} of the fp32 peak |")
        c = v.get("cpu_baseline")
        if c:
            print(f"\nCPU baseline on that line: {c['value']:.0f} residue·step/s at {c['cores']} threads ({c['all_physical_cores']['value']:.0f} at all {c['all_physical_cores']['cores']} physical cores, {c['single_thread']['value']:.0f} at one).\n")
