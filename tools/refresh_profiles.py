#!/usr/bin/env python
"""Turn gpurun_out/final/* (tools/collect_profiles.sh) into the committed artefacts of the round under profiles/ (r04_*)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, P = os.path.join(ROOT, "gpurun_out", "final") + "/", os.path.join(ROOT, "profiles") + "/"


def table(fn):
    out, sec = {}, None
    for line in open(d + fn):
        if line.startswith("## "):
            sec = line[3:].strip()
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \|", line)
        if m and sec:
            out.setdefault(sec, {})[m.group(1)] = float(m.group(3))
    return out


F, W, M = table("pmc_FETCH.md")["FETCH_SIZE"], table("pmc_WRITE.md")["WRITE_SIZE"], table("pmc_MFMA.md")
gui, busy = M["GRBM_GUI_ACTIVE"], M["SQ_VALU_MFMA_BUSY_CYCLES"]
kst = {}
for line in open(d + "kernel_stats.md"):
    m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m:
        kst[m.group(1)] = float(m.group(4))


def key(tab, prefix):
    return next(k for k in tab if k.startswith(prefix))


ET, A3, OP, EE = "edge_transition4_flat_kernel", "ipa_attn3_kernel", "opair_mfma_kernel", "edge_embed2_kernel"
util = {k: busy[k] / 1024 / (gui[k] / 8) for k in busy if k in gui}
et_f, et_w = F[ET] * 1024, W[ET] * 1024
B, N = 8, 300
z_bytes = B * N * N * 128 * 2
bias_bytes = B * 8 * 320 * 320 * 4 / 1.0  # fragment-order pair bias, Np = 320
alg_read, alg_write = z_bytes, z_bytes + B * 8 * N * N * 4
bench = json.load(open(d + "bench.json"))
et_us = kst[ET]

open(P + "r04_bench_c4_fp16_kernel_stats.md", "w").write(
    "# Round 4 — `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-reference-precision --no-all-samples` (1 x MI355X, config c4: fp16 mode, N=300, B=8)\n\n"
    "The command of the bench line in r04_bench.json without its two sub-records (fp32 `reference_precision`, 64-sample `all_samples_one_gpu`: the latter runs the same kernels at 8x the batch and would mix into the averages) (whole T = 500 trajectory: 5 warm-up steps on a scratch trajectory, priming\n"
    "forward + 500 steps timed).  Per-kernel totals over the whole process (prepare-time kernels and the D2H of the trajectories,\n"
    f"`__amd_rocclr_copyBuffer`, included).  edge_transition4_flat_kernel: {et_us:.1f} us average here vs {bench['roofline']['avg_launch_ms'] * 1e3:.1f} us\n"
    "from the HIP events of the bench's timed region.\n\n" + open(d + "kernel_stats.md").read())

if os.path.exists(d + "kernel_stats_c5.md"):
    c5 = json.load(open(d + "bench_c5.json"))
    open(P + "r04_bench_c5_fp32_kernel_stats.md", "w").write(
        "# Round 4 — `rocprofv3 --kernel-trace --stats -- python bench.py --config c5 --no-cpu-baseline` (1 x MI355X, config c5: fp32 mode, N = 1000, B = 4)\n\n"
        f"Bench line of the same configuration: {c5['value']:.0f} residue*step/s, {c5['ms_per_step']:.1f} ms per step; `edge_transition_f32ws_kernel` "
        f"{c5['roofline']['avg_launch_ms']:.2f} ms per launch = {c5['roofline']['frac'] * 100:.1f} % of the 157.3 TFLOP/s fp32 matrix peak (`v_mfma_f32_32x32x2_f32`; "
        "round 1: 51 ms = 34 %).\n\n" + open(d + "kernel_stats_c5.md").read())

def frame_row(prefix, what, alg_bytes_per_res):
    k = next((k for k in kst if k.startswith(prefix)), None)
    if k is None:
        return ""
    fk, wk = next((x for x in F if x.startswith(prefix)), None), next((x for x in W if x.startswith(prefix)), None)
    if fk and wk:
        nbytes, src = 2 * F[fk] * 1024 + W[wk] * 1024, "PMC"
    else:
        nbytes, src = alg_bytes_per_res * B * N, "algorithmic"
    gbs = nbytes / kst[k] / 1e3
    return f"| `{k[:40]}` | {what} | {nbytes / 1e6:.2f} MB ({src}) | {kst[k]:.1f} | {gbs:.0f} | {gbs / 8000 * 100:.1f} % |\n"


frame_rows = (frame_row("reverse_step_kernel", "SE(3) reverse step + atom37 of x_{t-1} + trajectory rows", 28 + 24 + 12 + 48 + 8 + 28 + 444 + 12)
              + frame_row("backbone_kernel", "atom37 / atom14 from frames + psi", 28 + 8 + 4 + 444 + 168)
              + frame_row("rot_score_kernel", "IGSO(3) score, R^3 score, tensor_7 / psi epilogue, last torsion layer", 1024 + 28 + 28 + 24 + 12 + 8)
              + frame_row("points16_kernel", "Rigid.apply of the q / k / v points -> MFMA fragment images (x4 per step)", 8 * 28 * 3 * 4 * 2)
              + frame_row("build_feats_kernel", "x_t split, node / pair feature rows", 1024))
fetch_ee = next((f"2 x {F[k] * 1024 / 1e6:.0f} MB" for k in F if k.startswith(EE)), "below the ten largest of the step")
hdr = f"""# Round 4 — PMC counters of the bench command (MI355X, config c4: fp16 mode, N=300, B=8)

Separate passes as MI355X_MICROARCH.md prescribes (never combined with sys/hip tracing; tools/collect_profiles.sh):
`rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline`
(three passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE).  FETCH / WRITE are in KiB per dispatch.
gfx950 correction (guide, HBM section): read bytes = 2 x FETCH_SIZE x 1024 for wide (16 B per lane) streaming reads; WRITE_SIZE x 1024
calibrates as is.  Both calibrations re-checked on this run:
* `opair_mfma_kernel` reads z exactly once (8 x 300^2 x 256 B = 184.3 MB) + the fp16 attention weights (12.3 MB) = 196.6 MB;
  2 x FETCH = {2 * F[key(F, OP)] * 1024 / 1e6:.1f} MB.  The x2 correction is exact for this access pattern.
* `edge_embed2_kernel` writes z (184.3 MB) + the fragment-order pair bias of block 0 (8 heads x 8 x 320^2 x 4 B = 26.2 MB, of which the
  padded-key slots are not written: 23.0 MB) = 207.3 MB; WRITE = {W[key(W, EE)] * 1024 / 1e6:.1f} MB.

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) (GRBM_GUI_ACTIVE is summed over the 8 XCDs).

## edge_transition4_flat_kernel (the roofline kernel of bench.py), per launch
* SQ_VALU_MFMA_BUSY_CYCLES {busy[ET]:,.0f} = 22,500 wave patches x 536 MFMAs x 32 cycles (v_mfma_f32_32x32x16_f16 at the bf16 rate).
* MFMA utilisation {util[ET] * 100:.1f} % of the launch's cycles (executed FLOPs 548,864 per pair; on the reference count of 688,128 per pair:
  {bench['roofline']['frac'] * 100:.1f} % of the 2.5 PFLOP/s peak at the bench's {bench['roofline']['avg_launch_ms'] * 1e3:.1f} us per launch).
* WRITE {et_w / 1e6:.1f} MB against 207.3 MB algorithmic (z' 184.3 MB + pair bias of the next block 23.0 MB): +{(et_w / 207.36e6 - 1) * 100:.0f} %, no spill traffic.
* FETCH: 2 x {et_f / 1e6:.1f} = {2 * et_f / 1e6:.1f} MB against 184.3 MB of z.  The x2 correction is right (see the o_pair calibration above), so
  {2 * et_f / 1e6 - 184.3:.0f} MB per launch are real memory-side requests beyond z: the 512 KB weight stream of a 256-pair tile is consumed by
  every CU once per tile (2813 tiles), and the z stream that passes through each 4 MB XCD L2 between two uses (32 CUs x (128 KB in +
  145 KB out) = 8.7 MB per tile round) evicts it, so every XCD re-fetches the stream ~{(2 * et_f / 1e6 - 184.3) / 8 / 0.524:.0f} times per launch from the
  Infinity Cache (these requests are counted by FETCH_SIZE although they never reach HBM: guide, "Infinity-Cache hits appear to be
  counted").  The fold-row images (2.5 MB) are the rest.  HBM-side the launch moves the algorithmic 184 MB in / 207 MB out.
* `roofline.traffic`: read {2 * et_f / 1e6:.0f} MB (memory-side, incl. {2 * et_f / 1e6 - 184.3:.0f} MB of Infinity-Cache hits) + write {et_w / 1e6:.0f} MB = {(2 * et_f + et_w) / 1e6:.0f} MB vs
  {(alg_read + alg_write) / 1e6:.0f} MB algorithmic.

## ipa_attn3_kernel (IPA attention: QK^T + pair bias + point distances, softmax, P V, o_pt) — the north star's "MFMA utilisation on the
attention GEMMs"
* {kst[key(kst, A3)]:.1f} us per call (x4 per step); SQ_VALU_MFMA_BUSY_CYCLES {busy[key(busy, A3)]:,.0f}; MFMA utilisation
  **{util[key(util, A3)] * 100:.1f} %** (round 3: 13.2 % at 55 us with fourteen fp32 MFMAs per key tile for the point logits; round 4 runs them as
  six fp16 hi / lo MFMAs from a key-point fragment image).  Bound by neither the matrix cores nor a memory path: the SQ counters
  (profiles/r04_pmc_sq_c4.md) show its waves parked on memory waits 51 % and issue-stalled 31 % of their cycles with the VALU 15 % busy —
  a per-CU throughput limit of dependent load -> MFMA -> softmax chains at 12 waves per CU (25.9 us at four samples, 46 at eight, 349
  at 64: linear in the block count).  640 blocks (64 (sample, head) x 10 query tiles, three per CU, one round) each pull K + V_hi +
  V_lo of their sample (shared by its eight heads, L2-resident) + bias 40 KB: 34 GB/s per CU, a quarter of what a CU's L2 path
  delivers.  Variants that stage K / V through LDS once per 64 - 128 queries (tools/micro/attention4_experiment.hip) measured slower
  at eight samples: 80 - 160 blocks leave most CUs idle.
* memory side: 2 x FETCH {2 * F[key(F, A3)] * 1024 / 1e6:.0f} MB, WRITE {W[key(W, A3)] * 1024 / 1e6:.0f} MB per call.

## HBM-bound passes
* `opair_mfma_kernel`: 196.8 MB in {kst[key(kst, OP)]:.1f} us = {2 * F[key(F, OP)] * 1024 / kst[key(kst, OP)] / 1e6:.2f} TB/s ({2 * F[key(F, OP)] * 1024 / kst[key(kst, OP)] / 1e6 / 8 * 100:.0f} % of the 8 TB/s peak, {2 * F[key(F, OP)] * 1024 / kst[key(kst, OP)] / 1e6 / 6.3 * 100:.0f} % of the 6.3 TB/s achievable).
* `edge_embed2_kernel`: 207 MB written in {kst[key(kst, EE)]:.1f} us = {W[key(W, EE)] * 1024 / kst[key(kst, EE)] / 1e6:.2f} TB/s; MFMA utilisation {util[key(util, EE)] * 100:.1f} % (47 GF): bound by neither
  (since the row-walk decomposition of round 2 its table rows come out of L2 / LDS: FETCH_SIZE {fetch_ee}; phase profile in DESIGN.md section 4.3; SQ counters: VALU 47 %, LDS array 54 %, matrix pipe 31 % busy, waves issue-stalled 41 %).

## Frame ops (the north star's "achieved HBM GB/s on the frame ops against CDNA4 peak")
B N = 2400 residues per launch: a few hundred bytes per residue, so these launches are latency-bound (one residue per lane, float64 chains),
nowhere near the 8 TB/s HBM peak — which is why they are folded into as few launches as possible (DESIGN.md section 4.3).  Bytes =
memory-side traffic of the PMC passes where the kernel is in their tables (2 x FETCH + WRITE), else the algorithmic bytes per residue.
| kernel | what | bytes per launch | us per launch | achieved GB/s | of 8 TB/s |
|---|---|---|---|---|---|
{frame_rows}
## MFMA utilisation of every kernel with matrix work
| kernel | us per launch | MFMA busy cycles | utilisation |
|---|---|---|---|
""" + "".join(f"| `{k[:70]}` | {kst.get(k, float('nan')):.1f} | {busy[k]:,.0f} | {util[k] * 100:.1f} % |\n" for k in sorted(util, key=lambda k: -busy[k])) + "\n"
open(P + "r04_pmc_bench_c4_fp16.md", "w").write(hdr + open(d + "pmc_FETCH.md").read() + open(d + "pmc_WRITE.md").read() + open(d + "pmc_MFMA.md").read())

rec = {"kernel": ET, "workload": {"precision": "fp16", "n_res": N, "samples_per_gpu": B},
       "fetch_size_kib": F[ET], "write_size_kib": W[ET], "read_bytes": 2 * et_f, "write_bytes": et_w, "traffic_bytes": 2 * et_f + et_w,
       "algorithmic_bytes": alg_read + alg_write, "algorithmic_read_bytes": alg_read, "algorithmic_write_bytes": alg_write,
       "infinity_cache_hit_read_bytes_estimate": 2 * et_f - alg_read,
       "sq_valu_mfma_busy_cycles": busy[ET], "grbm_gui_active_sum_xcd": gui[ET], "mfma_utilisation": util[ET],
       "ipa_attn3_mfma_utilisation": util[key(util, A3)],
       "source": "profiles/r04_pmc_bench_c4_fp16.md (rocprofv3 --pmc, separate passes; gfx950 FETCH x2 correction, calibrated on o_pair)"}
json.dump(rec, open(P + "r04_pmc_edge_transition.json", "w"), indent=1)

lines = {}
for tag, fn in (("c4_fp16_n300_b8", "bench.json"), ("c2_fp16_n128_b8", "bench_c2.json"), ("c3_fp16_mixed_bucket_of_8_complexes", "bench_c3.json"),
                ("c3e_fp16_equal_length_n776_b8", "bench_c3e.json"), ("c3w_fp16_mixed_whole_range_700_850", "bench_c3w.json"), ("c4_fp16_b64", "bench_c4_b64.json"),
                ("c5_fp32_n1000_b4", "bench_c5.json"), ("c5_shape_in_fp16", "bench_c5_fp16.json"), ("c4_fp32", "bench_c4_fp32.json"),
                ("c4_fp16_without_split_operands", "bench_c4_nosplit.json"), ("c4_fp16_b24", "bench_c4_b24.json"),
                ("c4_fp16_two_sub_batch_streams", "bench_c4_streams2.json"), ("c3_fp16_two_sub_batch_streams", "bench_c3_streams2.json")):
    if os.path.exists(d + fn):
        lines[tag] = json.load(open(d + fn))
        if tag.startswith("c4_fp16_n300") and lines[tag]["roofline"].get("traffic") is None:
            lines[tag]["roofline"]["traffic"] = {"total": rec["traffic_bytes"], "read": rec["read_bytes"], "write": rec["write_bytes"],
                                                 "algorithmic": rec["algorithmic_bytes"], "source": "profiles/r04_pmc_edge_transition.json"}
json.dump(lines, open(P + "r04_bench.json", "w"), indent=1)
print({k: (round(v["value"]), round(v["ms_per_step"], 3), round(v["roofline"]["whole_forward_frac"], 3)) for k, v in lines.items()})
print("ET util", round(util[ET], 4), "attn3 util", round(util[key(util, A3)], 4), "traffic MB", round(rec["traffic_bytes"] / 1e6))
