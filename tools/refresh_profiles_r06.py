#!/usr/bin/env python
"""Turn gpurun_out/final/* (tools/collect_profiles.sh) into the committed artefacts of round 6 under profiles/ (r06_*)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, P = os.path.join(ROOT, "gpurun_out", "final") + "/", os.path.join(ROOT, "profiles") + "/"
B, N = 8, 300


def table(fn):
    out, sec = {}, None
    for line in open(d + fn):
        if line.startswith("## "):
            sec = line[3:].strip()
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \|", line)
        if m and sec:
            out.setdefault(sec, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


def kstats(fn):
    out = {}
    for line in open(d + fn):
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            out[m.group(1)] = (int(m.group(2)), float(m.group(4)))
    return out


F, W, M = table("pmc_FETCH.md")["FETCH_SIZE"], table("pmc_WRITE.md")["WRITE_SIZE"], table("pmc_MFMA.md")
gui, busy = M["GRBM_GUI_ACTIVE"], M["SQ_VALU_MFMA_BUSY_CYCLES"]
kst = kstats("kernel_stats.md")
bench = json.load(open(d + "bench.json"))
drv = json.load(open(d + "bench_driver_cmd.json"))
util = {k: busy[k][1] / 1024 / (gui[k][1] / 8) for k in busy if k in gui}
ETA, ETB = "edge_transition4_flat_kernel<true, true>", "edge_transition4_flat_kernel<true, false>"  # z' stored (blocks 0, 1) / not stored (block 2)
pairs = B * N * N
z_b, bias_b, pz_b = pairs * 256, B * 8 * N * N * 4, pairs * 64
alg = {ETA: (z_b, z_b + bias_b + pz_b), ETB: (z_b, bias_b + pz_b)}


def first(tab, prefix):
    return next((k for k in tab if k.startswith(prefix)), None)


rows = []
traffic = {}
for k in (ETA, ETB):
    rd, wr = 2 * F[k][1] * 1024, W[k][1] * 1024
    traffic[k] = (rd, wr)
    rows.append(f"| `{k}` | {kst[k][1]:.1f} | {rd / 1e6:.1f} | {alg[k][0] / 1e6:.1f} | {wr / 1e6:.1f} | {alg[k][1] / 1e6:.1f} | {(rd + wr) / sum(alg[k]):.3f} | {util[k] * 100:.1f} % |")
per_launch = lambda f: (2 * f(ETA) + f(ETB)) / 3  # noqa: E731  (three launches per forward: two store z', the last one does not)
rd_l, wr_l = per_launch(lambda k: traffic[k][0]), per_launch(lambda k: traffic[k][1])
alg_r, alg_w = per_launch(lambda k: alg[k][0]), per_launch(lambda k: alg[k][1])
A3, OP, EE = first(kst, "ipa_attn3_kernel"), first(kst, "opair_pz_kernel"), first(kst, "edge_embed2_kernel")
hdr = f"""# Round 6 — PMC counters of the bench command (MI355X, config c4: fp16 mode, N=300, B=8)

Separate passes as MI355X_MICROARCH.md prescribes (never combined with sys / hip tracing; tools/collect_profiles.sh):
`rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python bench.py --steps 4 --warmup 1 --eager --no-cpu-baseline --no-reference-precision --no-all-samples`
(three passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE).  FETCH / WRITE are in KiB per dispatch; gfx950 correction
(guide, HBM section): read bytes = 2 x FETCH_SIZE x 1024 for wide (16 B per lane) streaming reads, WRITE_SIZE x 1024 as is.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).

## EdgeTransition launches (the roofline kernel of bench.py): two per forward store z' (`<true, true>`), the last one does not (`<true, false>`)
Algorithmic bytes per launch: read z {z_b / 1e6:.1f} MB; write z' {z_b / 1e6:.1f} MB (first two launches) + the next block's pair bias {bias_b / 1e6:.1f} MB
+ its pair_z image {pz_b / 1e6:.1f} MB (round 6).
| kernel | us (rocprofv3, bench.py whole trajectory) | read MB (2 x FETCH) | algorithmic read MB | write MB | algorithmic write MB | traffic / algorithmic | MFMA busy |
|---|---|---|---|---|---|---|---|
""" + "\n".join(rows) + f"""

Average over the three launches of a forward (`roofline.traffic` of bench.py): read {rd_l / 1e6:.1f} MB + write {wr_l / 1e6:.1f} MB = {(rd_l + wr_l) / 1e6:.1f} MB
against {(alg_r + alg_w) / 1e6:.1f} MB algorithmic ({(rd_l + wr_l) / (alg_r + alg_w):.3f} x; the excess reads are the 512 KB weight stream re-fetched from the
Infinity Cache behind the z stream, round 5: profiles/r05_et_zpol.md).

## Other kernels of the step
* `{A3}`: {kst[A3][1]:.1f} us per call, MFMA busy {util[first(util, 'ipa_attn3_kernel')] * 100:.1f} %, read {2 * F[first(F, 'ipa_attn3_kernel')][1] * 1024 / 1e6:.0f} MB, write {W[first(W, 'ipa_attn3_kernel')][1] * 1024 / 1e6:.0f} MB per call.
* `{OP}` (o_pair from the pair_z image, round 6): {kst[OP][1]:.1f} us per call, read {2 * F[first(F, 'opair_pz_kernel')][1] * 1024 / 1e6:.1f} MB
  (algorithmic: pair_z {pz_b / 1e6:.1f} MB + attention weights {B * N * 8 * 320 * 2 / 1e6:.1f} MB) = {2 * F[first(F, 'opair_pz_kernel')][1] * 1024 / kst[OP][1] / 1e6:.2f} TB/s.
* `{EE}`: {kst[EE][1]:.1f} us, write {W[first(W, 'edge_embed2_kernel')][1] * 1024 / 1e6:.1f} MB (z {z_b / 1e6:.1f} + pair bias {bias_b / 1e6:.1f} + pair_z {pz_b / 1e6:.1f}), MFMA busy {util[first(util, 'edge_embed2_kernel')] * 100:.1f} %.

## MFMA utilisation of every kernel with matrix work
| kernel | us per launch | MFMA busy cycles | utilisation |
|---|---|---|---|
""" + "".join(f"| `{k[:70]}` | {kst.get(k, (0, float('nan')))[1]:.1f} | {busy[k][1]:,.0f} | {util[k] * 100:.1f} % |\n" for k in sorted(util, key=lambda k: -busy[k][1])) + "\n"
open(P + "r06_pmc_bench_c4_fp16.md", "w").write(hdr + open(d + "pmc_FETCH.md").read() + open(d + "pmc_WRITE.md").read() + open(d + "pmc_MFMA.md").read())

rec = {"kernel": "edge_transition4_flat_kernel (average of a forward's three launches: two <true, true>, one <true, false>)",
       "workload": {"precision": "fp16", "n_res": N, "samples_per_gpu": B},
       "read_bytes": rd_l, "write_bytes": wr_l, "traffic_bytes": rd_l + wr_l, "algorithmic_bytes": alg_r + alg_w,
       "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w,
       "per_kernel": {k: {"fetch_size_kib": F[k][1], "write_size_kib": W[k][1], "read_bytes": traffic[k][0], "write_bytes": traffic[k][1],
                          "algorithmic_read_bytes": alg[k][0], "algorithmic_write_bytes": alg[k][1], "mfma_utilisation": util[k]} for k in (ETA, ETB)},
       "mfma_utilisation": per_launch(lambda k: util[k]), "ipa_attn3_mfma_utilisation": util[first(util, "ipa_attn3_kernel")],
       "source": "profiles/r06_pmc_bench_c4_fp16.md (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, separate passes, "
                 "bench.py --steps 4 --warmup 1 --eager; gfx950 FETCH x2 correction)", "round": 6}
json.dump(rec, open(P + "r06_pmc_edge_transition.json", "w"), indent=1)

et_evt = bench["roofline"]["avg_launch_ms"] * 1e3
et_prof = (2 * kst[ETA][1] + kst[ETB][1]) / 3
open(P + "r06_bench_c4_fp16_kernel_stats.md", "w").write(
    "# Round 6 — `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-reference-precision --no-all-samples` (1 x MI355X, config c4: fp16 mode, N=300, B=8)\n\n"
    "Whole T = 500 trajectory through the default product path (step-graph replays; the kernels inside a replay are traced like any other dispatch).  Per-kernel totals over the\n"
    "whole process (prepare-time kernels and the D2H of the trajectories, `__amd_rocclr_copyBuffer`, included).  EdgeTransition: two launches per forward store z'\n"
    f"(`<true, true>`), the last one does not (`<true, false>`): {kst[ETA][1]:.1f} / {kst[ETB][1]:.1f} us, {et_prof:.1f} us on average here vs {et_evt:.1f} us from the HIP events of the same\n"
    f"lease's bench line without the profiler ({bench['ms_per_step']:.4f} ms per step; the driver's command `--steps 20 --warmup 5` on the same lease: {drv['ms_per_step']:.4f} ms per step,\n"
    f"{drv['value']:,.0f} residue*step/s).\n\n" + open(d + "kernel_stats.md").read())
if os.path.exists(d + "kernel_stats_b64.md"):
    b64 = json.load(open(d + "bench_c4_b64.json"))
    open(P + "r06_bench_c4_fp16_b64_kernel_stats.md", "w").write(
        "# Round 6 — `rocprofv3 --kernel-trace --stats -- python bench.py --samples-per-gpu 64 --steps 18 --warmup 2 --no-cpu-baseline --no-reference-precision --no-all-samples`\n\n"
        f"All 64 samples of BASELINE configs[3] on one MI355X (the workload of `all_samples_one_gpu`).  Bench line of the same lease (40 steps): {b64['value']:,.0f} residue*step/s,\n"
        f"{b64['ms_per_step']:.2f} ms per step, whole forward {b64['roofline']['whole_forward_frac']:.3f} of the MFMA peak.\n\n" + open(d + "kernel_stats_b64.md").read())
if os.path.exists(d + "kernel_stats_c5.md"):
    c5 = json.load(open(d + "bench_c5.json"))
    open(P + "r06_bench_c5_fp32_kernel_stats.md", "w").write(
        "# Round 6 — `rocprofv3 --kernel-trace --stats -- python bench.py --config c5 --no-cpu-baseline` (1 x MI355X, config c5: fp32 mode, N = 1000, B = 4)\n\n"
        f"Bench line of the same configuration: {c5['value']:.0f} residue*step/s, {c5['ms_per_step']:.1f} ms per step; `edge_transition_f32ws_kernel` "
        f"{c5['roofline']['avg_launch_ms']:.2f} ms per launch = {c5['roofline']['frac'] * 100:.1f} % of the 157.3 TFLOP/s fp32 matrix peak.  (The fp32 kernels are unchanged in round 6.)\n\n"
        + open(d + "kernel_stats_c5.md").read())

lines = {}
for tag, fn in (("c4_fp16_n300_b8_driver_command", "bench_driver_cmd.json"), ("c4_fp16_n300_b8_whole_trajectory", "bench.json"), ("c2_fp16_n128_b8", "bench_c2.json"),
                ("c3_fp16_mixed_bucket_of_8_complexes", "bench_c3.json"), ("c3e_fp16_equal_length_n776_b8", "bench_c3e.json"),
                ("c3w_fp16_mixed_whole_range_700_850", "bench_c3w.json"), ("c4_fp16_b64", "bench_c4_b64.json"), ("c5_fp32_n1000_b4", "bench_c5.json"),
                ("c5_shape_in_fp16", "bench_c5_fp16.json"), ("c4_fp32", "bench_c4_fp32.json"), ("c4_fp16_without_split_operands", "bench_c4_nosplit.json"),
                ("c4_fp16_b24", "bench_c4_b24.json"), ("c4_fp16_eager", "bench_c4_eager.json")):
    if os.path.exists(d + fn) and os.path.getsize(d + fn):
        lines[tag] = json.load(open(d + fn))
json.dump(lines, open(P + "r06_bench.json", "w"), indent=1)
for k, v in lines.items():
    print(f"{k:42s} {v['value']:12.0f} {v['ms_per_step']:8.3f} ET {v['roofline']['avg_launch_ms']:.4f} ms frac {v['roofline']['frac']:.3f} wf {v['roofline']['whole_forward_frac']:.3f}"
          + (f" all64 {v['all64_value']:.0f} {v['all64_whole_forward_frac']:.3f}" if "all64_value" in v else "") + (f" fp32 {v['fp32_value']:.0f}" if "fp32_value" in v else ""))
print("traffic MB", round((rd_l + wr_l) / 1e6, 1), "algorithmic", round((alg_r + alg_w) / 1e6, 1), "ET util", round(rec["mfma_utilisation"], 4))
