#!/usr/bin/env python
"""Turn gpurun_out/final/* (tools/collect_profiles.sh) into the committed artefacts under profiles/."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, P = os.path.join(ROOT, "gpurun_out", "final") + "/", os.path.join(ROOT, "profiles") + "/"
def ctr(fn, name, kernel="edge_transition4_kernel"):
    sec = None
    for l in open(d + fn):
        if l.startswith("## "): sec = l[3:].strip()
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \|", l)
        if m and sec == name and m.group(1).startswith(kernel): return float(m.group(3))
    raise KeyError(name)
fetch, write = ctr("pmc_FETCH.md", "FETCH_SIZE"), ctr("pmc_WRITE.md", "WRITE_SIZE")
gui, busy = ctr("pmc_MFMA.md", "GRBM_GUI_ACTIVE"), ctr("pmc_MFMA.md", "SQ_VALU_MFMA_BUSY_CYCLES")
open(P + "r01_final_bench_bf16_kernel_stats.md", "w").write(
    "# Round 1 (final) — `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 20` (1 x MI355X, bf16, N=300, B=8)\n\n"
    "26 forwards (5 warm-up + 20 timed + priming); per-kernel totals over the whole process (prepare-time kernels included).\n\n" + open(d + "kernel_stats.md").read())
hdr = f"""# Round 1 (final) — PMC counters of the bench command (MI355X, bf16, N=300, B=8)

Collected in separate passes as MI355X_MICROARCH.md prescribes (never combined with sys/hip tracing; tools/collect_profiles.sh):
`rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline`
(three passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE).  FETCH/WRITE are in KiB.
gfx950 correction: HBM read bytes = 2 x FETCH_SIZE x 1024 for wide coalesced reads; WRITE_SIZE x 1024 calibrates exactly
(edge_embed2_kernel writes 720,000 pair rows x 256 B = 184.3 MB of z plus 23 MB of pair bias: 203,818 KiB).

## edge_transition4_kernel (the roofline kernel of bench.py), per launch
* SQ_VALU_MFMA_BUSY_CYCLES {busy:,.0f} = 22,500 wave patches x 536 MFMAs x 32 cycles, exactly (edge_transition3: 463.68 M).
* GRBM_GUI_ACTIVE {gui:,.0f} summed over the 8 XCDs -> {gui/8/1e6:.3f} M cycles per launch; MFMA utilisation =
  {busy/1e6:.2f} M / (1024 SIMDs x {gui/8/1e6:.3f} M) = **{busy/1024/(gui/8)*100:.1f} %** of executed matrix cycles (edge_transition3: 42.8 %), and the
  executed cycles are 17 % fewer for the same result.
* WRITE_SIZE {write:,.1f} KiB = {write*1024/1e6:.1f} MB: the algorithmic bytes (z' 184.3 MB + pair bias of the next block 23.0 MB = 207.4 MB).
  FETCH_SIZE {fetch:,.1f} KiB -> 2 x = {2*fetch*1024/1e6:.1f} MB with the prescribed correction (z 184.3 MB + per-residue images and the
  weight stream, which hit L2 / MALL); `roofline.traffic` = 2 x FETCH + WRITE = **{(2*fetch+write)*1024/1e6:.0f} MB** vs 397.7 MB algorithmic.
* History of this number: the first edge_transition4 build had 18 spilled registers, and its counters were WRITE 293,012 /
  FETCH 208,423 KiB: a spilled dword is 256 B of scratch per wave and tile = 5.8 MB of HBM writes per launch (plus the
  reload), and a scratch reload or any other global load next to the inline-asm LDS-DMAs is a `vmcnt` wait that drains the
  whole DMA queue.  Removing the spills (lane index from v_mbcnt, scalar wave index, 32-bit image offsets) and the in-loop
  global loads (pair mask / linear_b bias from LDS) took WRITE to the algorithmic 207 MB and the launch from 0.343 to 0.30 ms.

"""
open(P + "r01_final_pmc_bench_bf16.md", "w").write(hdr + open(d + "pmc_FETCH.md").read() + open(d + "pmc_WRITE.md").read() + open(d + "pmc_MFMA.md").read())
traffic = (2 * fetch + write) * 1024
json.dump({"kernel": "edge_transition4_kernel", "workload": {"precision": "bf16", "n_res": 300, "samples_per_gpu": 8},
           "fetch_size_kib": fetch, "write_size_kib": write, "hbm_read_bytes": 2 * fetch * 1024, "hbm_write_bytes": write * 1024,
           "traffic_bytes": traffic, "sq_valu_mfma_busy_cycles": busy, "grbm_gui_active_sum_xcd": gui,
           "source": "profiles/r01_final_pmc_bench_bf16.md (rocprofv3 --pmc, separate passes, gfx950 FETCH x2 correction)"},
          open(P + "r01_pmc_edge_transition.json", "w"), indent=1)
lines = {k: json.loads(open(d + f).read()) for k, f in (("bf16_n300", "bench.json"), ("fp32_n300", "bench_fp32.json"), ("bf16_n128", "bench_n128.json"))}
for v in lines.values():  # the bench read the previous round's traffic file: store the line with the counters of THIS collection
    if v.get("roofline", {}).get("traffic"): v["roofline"]["traffic"] = traffic
json.dump(lines, open(P + "r01_final_bench.json", "w"), indent=1)
print({k: (round(v["value"]), round(v["ms_per_step"], 3)) for k, v in lines.items()}, "util", round(busy / 1024 / (gui / 8), 4), "traffic MB", round(traffic / 1e6))
