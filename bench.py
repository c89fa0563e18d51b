#!/usr/bin/env python
"""Headline benchmark: residue x diffusion-steps / s of the sampler hot loop (BASELINE.json metric).

    python bench.py --gpus 1 [--steps K] [--warmup W] [--config c4|c2|c3|c5] [--precision fp16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs, per GPU; weak scaling, samples sharded round-robin, no collective on the data path):
  c4 (default, the config the metric is quoted on): de novo, N = 300, 8 samples batched, T = 500, fp16 mode
  c2: de novo, N = 128, 8 samples, T = 500, fp16 mode
  c3: TCR-pMHC-like inpainting, 8 different synthetic 4-chain complexes of 700 - 850 residues (seq_idx gap 200, two CDR3-like
      windows) in ONE padded batch, T = 100, fp16 mode; value counts the real residues only (c3e: 8 samples of one 776-residue complex)
  c5: long-chain inpainting, N = 1000 (2 chains), 4 samples, one diffused window of 50, T = 100, fp32 mode
fp16 mode = fp16 MFMA operands / pair representation with split (hi + lo) operands on the node path: the mode whose per-step
backbone RMSD against the reference is < 1e-3 A (tests/test_gpu_sizes.py::test_teacher_forced_fp16_meets_the_north_star_bound).

A "step" = one reverse-diffusion step of the whole batch exactly as inference_fn runs it with aux_traj=True (the reference's
caller always passes it, experiments/inference.py:218,331): score-network forward incl. its backbone atoms, fused SE(3)
reverse step + atom37 frame of x_{t-1}, trajectory writes.  By default K = T: W warm-up steps on a scratch trajectory, then
the WHOLE trajectory is timed (self-conditioning priming forward + T steps, t = 1 -> min_t, the last step takes the x_0
branch); with K < T a window of K consecutive steps in the middle of the schedule is timed (priming untimed), issued exactly as
inference_fn issues them (chunked graph replays).  Inputs, weights and the noise tape are resident in HBM before the timed region; value = B*N*K*n_gpus / max-over-ranks.
Every sample draws x_T and its noise tape from its own stream (seed + global sample index, framedipt_amd/sharding.py).

Also on the JSON line:
  loop         : how the timed steps were issued — the default product path (inference.ReverseLoop: cursor-addressed launches replayed
                 as HIP graphs; `--eager`: launch by launch), host time per step, set-up and graph capture before the timed region.
  roofline     : the dominant kernel (EdgeTransition, 89 % of the reference FLOPs) timed with HIP events recorded by the library
                 on the launch stream around its launches in 4 (K <= 32) or 16 steps of the timed region — those steps are enqueued
                 launch by launch through the same step cursor (events inside a captured graph cannot be timed on ROCm) —, read back
                 after the region; achieved = reference-formulation FLOPs per launch / duration (executed FLOPs stated beside it).
  all_samples_one_gpu : c4 on one GPU only — all 64 samples of BASELINE configs[3] in one batch, a few steps, own roofline.
  (flat copies: whole_forward_frac, fp32_value / fp32_ms_per_step / fp32_whole_forward_frac, all64_value / ... for parsers that keep top-level keys)
  cpu_baseline : the oracle loop with a torch-CPU forward (oracle/torch_port.py) on this box's host cores: best thread count of a short scan,
                 all physical cores and one thread; bounded sample, rank 0, N = 1 only.
  reference_precision : the same workload in fp32 (the reference's arithmetic) for a few steps: value + roofline of that mode.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ET_FLOPS_PER_PAIR = 688128.0   # 2*(2*384^2 + 384*128): EdgeTransition, reference formulation
NOMINAL_GHZ = 2.4  # engine clock of PEAK_TFLOPS (MI355X_MICROARCH.md)
# edge_transition4: 504 MFMAs of 32x32x16 per 32-pair wave tile for the three layers (round 6: 536 before the z part of the residual
# trunk(x) + x was merged into the hidden features it shares its final-layer columns with) + 24 in the epilogue (8 for the next block's pair
# bias linear_b(z'), 16 for its pair_z = down_z(z') on hi + lo weights).  The reference-formulation count of the launch stays
# ET_FLOPS_PER_PAIR: the epilogue's products are other modules' work (2 * 128 * (8 + 32) = 10,240 FLOP per pair, FUSED_FLOPS_PER_PAIR)
# that rides on this launch; `frac` prices the launch on the EdgeTransition FLOPs alone, `frac_incl_fused` credits the fused projections
ET4_EXEC_FLOPS_PER_PAIR = (504 + 24) * 32 * 32 * 16 * 2 / 32.0
FUSED_FLOPS_PER_PAIR = 2.0 * 128 * (8 + 32)  # linear_b (ipa_pytorch.py:247) + down_z (:158,318) of the next block, reference formulation
PEAK_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md

CONFIGS = {
    "c2": dict(n=128, b=8, t=500, prec="fp16", inpaint=False),
    # c3: a mixed-length batch as BASELINE configs[2] would feed it — 62 synthetic 4-chain complexes, N ~ U[700, 850], grouped by
    # length into batches of 8 (framedipt_amd/sharding.py: batches_mixed); the bench times the MIDDLE batch (rank r: the r-th after
    # it): 8 DIFFERENT complexes, one sample each, padded to the longest with res_mask = 0 rows (stack_items_padded);
    # c3w: the worst case, 8 complexes spanning the whole length range in one batch;
    # c3e: the same number of samples of ONE complex of the mean length (the equal-N reference for the padding overhead)
    "c3": dict(n=None, b=8, t=100, prec="fp16", inpaint=True, mixed=(700, 850, 62), windows=((92, 106), (330, 345))),
    "c3w": dict(n=None, b=8, t=100, prec="fp16", inpaint=True, mixed=(700, 850, 0), windows=((92, 106), (330, 345))),
    "c3e": dict(n=776, b=8, t=100, prec="fp16", inpaint=True, chains=(214, 258, 9, 295), windows=((92, 106), (330, 345))),
    "c4": dict(n=300, b=8, t=500, prec="fp16", inpaint=False),
    "c5": dict(n=1000, b=4, t=100, prec="fp32", inpaint=True, chains=(500, 500), windows=((40, 90),)),
}


def pmc_traffic(precision: str, n: int, b: int):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (profiles/), only when the bench
    runs the profiled configuration; the counters cannot be read from inside this process."""
    for name in ("r06_pmc_edge_transition.json", "r05_pmc_edge_transition.json", "r04_pmc_edge_transition.json", "r03_pmc_edge_transition.json", "r02_pmc_edge_transition.json", "r01_pmc_edge_transition.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            w = rec["workload"]
            if (w["precision"], w["n_res"], w["samples_per_gpu"]) == (precision, n, b):
                return {"total": rec["traffic_bytes"], "read": rec.get("read_bytes"), "write": rec.get("write_bytes"),
                        "algorithmic": rec.get("algorithmic_bytes"), "source": "profiles/" + name}
        except (OSError, KeyError, ValueError):
            pass
    return None


def flops_per_forward(n: int, inpainting: bool = False) -> float:
    """SURVEY.md 8d / BASELINE.md section 3: reference-formulation FLOPs of one score-network forward."""
    c2 = 2248960.0 + (107008.0 - 96256.0 if inpainting else 0.0)
    return c2 * n * n + 32.4e6 * n


def synthetic_complex(chain_lens, windows, seed=0):
    """Per-structure feature dict of a synthetic multi-chain complex (what data_utils.process_csv_row would hand over)."""
    rng = np.random.default_rng(seed)
    n = int(sum(chain_lens))
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    tr = np.cumsum(rng.standard_normal((n, 3)) * 2.2, 0)
    tr -= tr.mean(0)
    dm = np.zeros(n)
    for a, b in windows:
        dm[a:b] = 1
    seq_idx, chain_idx, off = [], [], 0
    for c, L in enumerate(chain_lens):
        seq_idx.append(np.arange(L) + off + c * 200)
        chain_idx.append(np.full(L, c))
        off += L
    tors = rng.standard_normal((n, 7, 2))
    tors /= np.linalg.norm(tors, axis=-1, keepdims=True)
    return {"rigids_0": np.concatenate([q, tr], -1).astype(np.float32), "diffuse_mask": dm, "aatype": rng.integers(0, 20, n),
            "seq_idx": np.concatenate(seq_idx), "chain_idx": np.concatenate(chain_idx), "torsion_angles_sin_cos": tors}


def cpu_baseline(n: int, conf, seed: int, steps: int = 10):
    """The oracle's loop (x_T + priming forward + `steps` reverse steps of the T = 500 schedule) with the score-network forward on
    torch-CPU ops (oracle/torch_port.py: the NumPy restatement's formulas on the multithreaded kernels the reference's own
    torch-CPU path uses), torch.set_num_threads(k) for k = the cores this process may run on, and one forward at k = 1."""
    import torch
    from framedipt_amd import weights as W
    from oracle import diffuser as od
    from oracle import inference as oi
    from oracle.torch_port import TorchScoreNetwork
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    odiff = od.SE3Diffuser(conf.diffuser)
    net = TorchScoreNetwork(conf.model, odiff, W.synth_state_dict(W.param_shapes(conf.model), seed), tables=tables)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    phys = max(1, cores // 2) if cores > 16 else cores  # SMT siblings do not add matrix throughput
    prev = torch.get_num_threads()
    feats = oi.unconditional_feats(odiff, n)
    tp = np.ones((1,), dtype=np.float32)
    sched = np.linspace(0.01, 1.0, 500)[::-1]
    feats = oi.set_t_feats(feats, sched[0], tp, odiff)

    def one_forward(k):
        torch.set_num_threads(k)
        net(feats)  # warm-up (thread pool, allocator)
        t = time.perf_counter()
        net(feats)
        return time.perf_counter() - t

    # torch-CPU does not scale to a whole 128-core host on this forward (measured on the GPU box: 2.9 s at 1 thread, 1.07 s at 16,
    # 1.8 s at 64, 3.7 s at 128): the baseline is quoted at the best of a short scan, the all-cores and single-thread figures beside it
    scan = {k: one_forward(k) for k in sorted({min(8, phys), min(16, phys), min(32, phys)})}
    threads = min(scan, key=scan.get)
    t_all = one_forward(phys) if phys not in scan else scan[phys]
    t_one = one_forward(1)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    feats["sc_ca_t"] = net(feats)["rigids"][..., 4:]
    for k in range(steps):
        feats, *_ = oi.one_step(net, odiff, feats, sched[k], 0.01, 1 / 500, tp, noise_scale=0.1)
    el = time.perf_counter() - t0
    fwd = steps + 1
    torch.set_num_threads(prev)
    per_step = el * (steps + 1 / 500) / fwd / steps  # priming forward amortised over the T = 500 steps of a trajectory
    return {"value": n / per_step, "unit": "residue*step/s", "cores": threads, "kind": "port",
            "sample": f"oracle loop with the torch-CPU forward (oracle/torch_port.py), de novo N={n}, B=1, {fwd} forwards + {steps} reverse "
                      f"steps of the T=500 schedule ({el:.1f} s wall at torch.set_num_threads({threads}), the best of {sorted(scan)} on {cores} logical CPUs; "
                      f"NOT an all-core figure: all {phys} physical cores are SLOWER on this forward, see all_physical_cores)",
            "all_physical_cores": {"value": n / (t_all * 501 / 500), "cores": phys, "sample": f"one forward at {phys} threads: {t_all:.2f} s"},
            "single_thread": {"value": n / (t_one * 501 / 500), "cores": 1, "sample": f"one forward at 1 thread: {t_one:.2f} s"},
            "note": "a baseline, not the target; the reference's own torch-CPU loop measured 213 residue*step/s at N=300 on 8 cores of "
                    "the build container (SURVEY.md section 6)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: the whole schedule of the config)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--n-res", type=int, default=None)
    ap.add_argument("--samples-per-gpu", type=int, default=None)
    ap.add_argument("--num-t", type=int, default=None)
    ap.add_argument("--precision", default=None, choices=["fp16", "fp32"])
    ap.add_argument("--kernel-flags", type=lambda s: int(s, 0), default=0, help="FdiptDims.kernel_flags (development)")
    ap.add_argument("--bf16", action="store_true", help="the half-precision mode with bf16 operands / pair representation (the -DFDIPT_HALF_BF16 "
                    "build of the library, lib/libfdipt_hip_bf16.so): what BASELINE configs[1] literally names; misses the parity bar (DESIGN.md section 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-precision", action="store_true", help="skip the fp32 (reference precision) sub-record")
    ap.add_argument("--reference-steps", type=int, default=6, help="timed steps of the fp32 sub-record")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak (default): --samples-per-gpu samples on EVERY GPU; strong: the "
                    "config's whole sample set (c4: BASELINE configs[3]'s 64 samples) split over the GPUs, so that value(N) / value(1) is the "
                    "speed-up of a fixed job")
    ap.add_argument("--total-samples", type=int, default=64, help="--scaling strong: samples of the whole job")
    ap.add_argument("--no-all-samples", action="store_true", help="skip the all_samples_one_gpu sub-record (c4, one GPU: configs[3]'s 64 samples "
                    "in one batch, a few steps)")
    ap.add_argument("--all-samples-steps", type=int, default=8, help="timed steps of the all_samples_one_gpu sub-record")
    ap.add_argument("--reserve-cus", type=int, default=48, help="with --streams > 1: CUs the persistent pair kernels leave to the other streams")
    ap.add_argument("--main-cus", type=int, default=0, help="experiment: run everything on a CU-masked stream with this many CUs (tools/cu_streams.py; HISTORY.md round 4)")
    ap.add_argument("--et-reserve", type=int, default=0, help="experiment: CUs the persistent pair kernels leave free (single stream)")
    ap.add_argument("--eager", action="store_true", help="time the launch-by-launch loop (inference_fn(graph=False)) instead of the default "
                    "step-graph replays: the host-enqueue-bound comparison line")
    ap.add_argument("--graph-chunk", type=int, default=None, help="steps per replay of the chunk graph (inference.ReverseLoop.GRAPH_CHUNK)")
    ap.add_argument("--streams", type=int, default=1, help="sub-batches on this many HIP streams (same results; not the default: the "
                    "roofline kernel's launches are then sub-batch sized and rocprofv3 serialises the streams)")
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    if a.n_res is not None:
        if cfg["inpaint"]:
            raise SystemExit("--n-res applies to the de novo configs")
        cfg["n"] = a.n_res
    B = a.samples_per_gpu or cfg["b"]
    N = cfg["n"]  # (mixed-length workload: set to the padded length below)
    T, prec = a.num_t or cfg["t"], a.precision or cfg["prec"]
    if a.bf16:
        if prec != "fp16":
            raise SystemExit("--bf16 replaces the half-precision mode")
        prec = "bf16"
        os.environ["FDIPT_LIB"] = os.path.join(ROOT, "framedipt_amd", "lib", "libfdipt_hip_bf16.so")  # (read when the package loads the library)
    K = a.steps if a.steps is not None else T
    if K > T or K < 1:
        raise SystemExit("--steps must lie in [1, num_t]")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # FDIPT_BENCH_ONE_GPU=1 (tests only): all ranks share GPU 0 and rendezvous over gloo - the multi-rank code path (sharding of the
    # samples, barriers, max-over-ranks timing, rank-0 line) on a one-GPU box, where RCCL refuses two ranks on one device
    one_gpu = os.environ.get("FDIPT_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if a.main_cus:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import cu_streams  # experiment helper (tools/cu_streams.py): CU-masked HIP streams
        n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
        torch.cuda.set_stream(cu_streams.pair(dev, n_cu - a.main_cus).main)
    n_ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_gpu else "nccl", rank=rank, world_size=world)
        # what the process group itself saw (the SCALE record can answer "did RCCL connect N ranks"): every rank contributes a one
        ones = torch.ones(1, device="cpu" if one_gpu else dev, dtype=torch.int32)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
        if n_ranks_seen != world or dist.get_world_size() != world:
            raise SystemExit(f"process group saw {n_ranks_seen} ranks (world size {dist.get_world_size()}), expected {world}")

    from framedipt_amd import _lib, config, inference, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import ConditionalSampler, UnconditionalSampler

    lib = _lib.load()
    if a.graph_chunk is not None:
        inference.ReverseLoop.GRAPH_CHUNK = a.graph_chunk
    inp = cfg["inpaint"]
    conf = config.base_config(inpainting=inp)
    diff = SE3Diffuser(conf.diffuser, device=dev)
    net = ScoreNetwork(conf.model, diff, inpainting=inp, precision=prec, kernel_flags=a.kernel_flags).load_synthetic(7).to(dev)
    if a.scaling == "strong":
        if a.total_samples % world:
            raise SystemExit(f"--scaling strong: {a.total_samples} samples do not split evenly over {world} GPUs")
        B = a.total_samples // world
    n_total = B * world
    mixed = cfg.get("mixed")
    if mixed:
        # n_total different complexes: total length ~ U[lo, hi] (fixed seed), chains in the proportions 200 : 240 : 9 : 275
        rng = np.random.default_rng(2024)
        if mixed[2]:  # the lengths of the whole set, bucketed as run_sharded would; this job's ranks take consecutive middle buckets
            pool = [int(v) for v in rng.integers(mixed[0], mixed[1] + 1, size=mixed[2])]
            buckets = [g for g in sharding.batches_mixed(pool, B) if len(g) == B]
            picks = [pool[p] for r in range(world) for p in buckets[(len(buckets) // 2 + r) % len(buckets)]]
        else:
            picks = [int(v) for v in rng.integers(mixed[0], mixed[1] + 1, size=n_total)]
        structs = []
        for k in range(n_total):
            n_k = picks[k]
            c = [int(round(n_k * f)) for f in (200 / 724, 240 / 724)] + [9]
            c.append(n_k - sum(c))
            structs.append((f"synthetic{k}", synthetic_complex(tuple(c), cfg["windows"], seed=k)))
        ds = ConditionalSampler.from_features(structs, diff, dev, samples=1)
    elif inp:
        ds = ConditionalSampler.from_features([("synthetic", synthetic_complex(cfg["chains"], cfg["windows"]))], diff, dev,
                                              samples=n_total)
    else:
        ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1,
                                                  "samples_per_length": n_total}), diff, dev)
    # independent samples: global sample index -> rank (round-robin), per-sample stream seed + index
    mine = list(range(rank * B, (rank + 1) * B)) if mixed else sharding.shard_indices(n_total, rank, world)  # (mixed: a bucket per rank)
    items = [sharding.seeded_item(ds, i, a.seed, diff, T, 0.01) for i in mine]
    if mixed:
        feats, tape, lengths = sharding.stack_items_padded(items)
        N = int(feats["rigids_t"].shape[1])
    else:
        feats, tape = sharding.stack_items(items)
        lengths = [N] * B
    # real residues of all ranks' batches (the padded rows are not counted) and their reference-formulation FLOPs per forward
    if world > 1:
        all_len = [None] * world
        dist.all_gather_object(all_len, lengths)
        all_len = [n for part in all_len for n in part]
    else:
        all_len = lengths
    res_per_step = int(sum(all_len))
    fwd_flops_per_step = float(sum(flops_per_forward(n, inp) for n in all_len))

    nb = conf.model.ipa.num_blocks
    n_ev = nb - 1

    def timed_region(net, prec, K, warmup, streams, feats=feats, tape=tape, B=B):
        """W warm-up steps on a scratch trajectory, then K timed steps (the whole trajectory incl. the priming forward when K == T)
        of the batch (`feats`, `tape`: the rank's batch unless given) with `net`; returns (seconds, D2H seconds, D2H bytes,
        EdgeTransition launch ms list, clock counters, B_ev)."""
        def new_loop():
            if streams > 1:
                return inference.StreamedLoops(net, diff, feats, streams, T, 0.01, noise_tape=tape, reserve_cus=a.reserve_cus, aux_traj=True,
                                               noise_scale=0.1, inpainting=inp, experimental=True)
            lp = inference.ReverseLoop(net, diff, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, inpainting=inp,
                                       noise_tape=tape)
            lp.st.reserve_cus = a.et_reserve
            return lp
        # timed steps: the whole schedule, or a window of K CONSECUTIVE steps in the middle of it (the loop's steady state: chunked graph
        # replays exactly as inference_fn issues them; the window starts from x_T's frames copied into its first row, so every timed step
        # works on a valid state and hands it to the next one)
        k0 = 0 if K == T else max(0, (T - 1 - K) // 2)
        steps = list(range(k0, k0 + K))
        # HIP events around every EdgeTransition launch of a few of the timed steps (recorded on the launch stream by the library):
        # up to 16 of a whole trajectory, at most 4 when K <= 32
        n_samp = min(K, 16 if K > 32 else 4)
        sampled = sorted({steps[int(round(i * (K - 1) / max(n_samp - 1, 1)))] for i in range(n_samp)})
        events = {}
        for k in sampled:
            ev_s, ev_e = (C.c_void_p * nb)(), (C.c_void_p * nb)()
            for i in range(n_ev):
                for arr in (ev_s, ev_e):
                    h = C.c_void_p()
                    _lib.check(lib.fdipt_event_create(C.byref(h)))
                    arr[i] = h
            events[k] = (ev_s, ev_e)
        warm = new_loop()  # scratch trajectory: touches every buffer / code object once
        warm.prime()
        for k in range(min(warmup, T)):
            warm.step(k)
        torch.cuda.synchronize()
        del warm
        t_setup = time.perf_counter()
        loop = new_loop()
        st = loop.st
        # in-kernel shader-clock probe of the EdgeTransition launches: opt-in, into a caller-owned buffer (FdiptForwardArgs.clock_out),
        # only on the sampled steps the HIP events bracket
        clk_dev = torch.zeros(3, dtype=torch.int64, device=dev)
        # the product path (inference.ReverseLoop, graph=True): steps are replays of a HIP graph captured once per trajectory; the few
        # event-bracketed steps are enqueued launch by launch through the same step cursor (HIP events recorded inside a graph cannot
        # be timed on ROCm).  Captures happen here, before the timed region (nothing runs); their cost is reported as graph_capture_ms
        graphed = streams == 1 and loop.graph and not a.eager
        if graphed:
            loop.prepare()
        elif streams == 1:
            loop.graph = False
        setup_s = time.perf_counter() - t_setup

        def bracket(k):
            st.ev_start, st.ev_stop = events.get(k, (None, None))
            st.clock_out = clk_dev if k in events else None
        if K < T:
            loop.prime()
            for lp in (loop.loops if streams > 1 else [loop]):
                lp.rigid_traj[k0].copy_(lp.rigid_traj[0])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if streams == 1:
            if K == T:
                loop.prime()
            loop.run_steps(k0, k0 + K, eager_steps=sampled, before_step=bracket)
        else:
            if K == T:
                loop.prime()
            for k in steps:
                bracket(k)
                loop.step(k)
        host_s = time.perf_counter() - t0  # host time to enqueue the region (the GPU runs behind it)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st.ev_start = st.ev_stop = st.clock_out = None
        clk = [int(v) for v in clk_dev.cpu()]
        d2h = d2h_bytes = None
        if K == T:  # D2H of the whole trajectories, as inference_fn returns them (never part of `value`; only meaningful for a whole run)
            t1 = time.perf_counter()
            res = loop.results()
            d2h = time.perf_counter() - t1
            d2h_bytes = sum(v.nbytes for v in res.values() if hasattr(v, "nbytes"))
            del res
        et_ms = []
        for k in sampled:
            for i in range(n_ev):
                ms = C.c_float()
                _lib.check(lib.fdipt_event_elapsed_ms(events[k][0][i], events[k][1][i], C.byref(ms)))
                et_ms.append(ms.value)
            for arr in events[k]:
                for i in range(n_ev):
                    lib.fdipt_event_destroy(arr[i])
        if world > 1:
            tt = torch.tensor([el, d2h or 0.0], device="cpu" if one_gpu else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el, d2h = float(tt[0].item()), (float(tt[1].item()) if d2h is not None else None)
            dist.barrier()
        B_ev = loop.loops[0].B if streams > 1 else B  # samples in the launches the events bracket (first sub-batch)
        loop_info = argparse.Namespace(capture_seconds=getattr(loop, "capture_seconds", 0.0))
        del loop
        return el, d2h, d2h_bytes, et_ms, (clk[0], clk[1]), B_ev, {"graph": bool(graphed), "setup_ms": setup_s * 1e3, "host_enqueue_ms_per_step": host_s / K * 1e3,
                                                              "graph_capture_ms": getattr(loop_info, "capture_seconds", 0.0) * 1e3}

    def record(prec, K, el, et_ms, clk, B_ev, kernel_flags, res_per_step=res_per_step, fwd_flops_per_step=fwd_flops_per_step, B=B):
        """value / ms_per_step / roofline of one timed region."""
        res_steps = res_per_step * K
        value = res_steps / el
        et = float(np.mean(et_ms)) * 1e-3
        et_flops = ET_FLOPS_PER_PAIR * B_ev * N * N
        peak = PEAK_TFLOPS[prec]
        achieved = et_flops / et / 1e12
        fwd_per_step = (T + 1) / T if K == T else 1.0
        fwd_tflops = fwd_flops_per_step * K / el / world * fwd_per_step / 1e12  # whole-forward view, per GPU (real residues only)
        et4 = prec in ("fp16", "bf16") and N % 4 == 0 and not (kernel_flags & 1)  # (FDIPT_KF_ET3 forces the fallback kernel)
        ghz = clk[0] / clk[1] / 10 if clk[1] else None
        return value, {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": pmc_traffic(prec, N, B), "kernel": "edge_transition4_flat_kernel" if et4 else
            ("edge_transition3_kernel" if prec != "fp32" else "edge_transition_f32ws_kernel"),
            "avg_launch_ms": et * 1e3, "launches_timed": len(et_ms), "flops_per_launch": et_flops,
            "executed_flops_per_launch": ET4_EXEC_FLOPS_PER_PAIR * B_ev * N * N if et4 else None,
            "executed_frac": ET4_EXEC_FLOPS_PER_PAIR * B_ev * N * N / et / 1e12 / peak if et4 else None,
            # shader clock the kernel's blocks actually ran at (s_memtime / s_memrealtime inside the kernel): power
            # management holds it below the 2.4 GHz of `peak`; frac_at_clock prices the same FLOPs against the matrix
            # peak at that clock
            "frac_incl_fused": (ET_FLOPS_PER_PAIR + FUSED_FLOPS_PER_PAIR) * B_ev * N * N / et / 1e12 / peak if et4 else None,
            "clock_ghz": ghz, "frac_at_clock": achieved / (peak * ghz / NOMINAL_GHZ) if ghz else None,
            "whole_forward_tflops": fwd_tflops, "whole_forward_frac": fwd_tflops / peak}

    PREC_MODE = {"fp16": "fp16 MFMA operands / pair representation (fp16 stands in for the bf16 BASELINE configs[1] names: same MFMA rate, "
                         "three more significand bits), split (hi+lo) operands on every per-residue product and the attention's P V, fp32 "
                         "accumulation / frames / statistics; per-step backbone RMSD vs the reference < 1e-3 A up to trained-weight scale bb_gain 0.3 "
                         "(0.8e-3 A worst step) and a MISS of that bar at bb_gain 0.5 (1.5e-3 A worst step: tests/test_gpu_round4.py asserts the "
                         "measured value); 14 % of the T=500 schedule (0.011 < t < 0.15, where the reference's own fp32 IGSO(3) series is round-off) "
                         "is checked by a property fence, not against the reference (tests/test_gpu_sizes.py)",
                 "bf16": "bf16 MFMA operands / pair representation (the -DFDIPT_HALF_BF16 build), split operands as in the fp16 mode; per-step "
                         "backbone RMSD vs the reference 1.7e-3 A worst / 4e-4 A median (teacher-forced, N=64, T=20: "
                         "tests/test_gpu_round4.py::test_bf16_build_per_step_numbers): outside the 1e-3 A parity bar, a comparison line only",
                 "fp32": "fp32 (v_mfma_f32_32x32x2_f32): the reference's arithmetic"}
    el, d2h, d2h_bytes, et_ms, clk, B_ev, loop_rec = timed_region(net, prec, K, a.warmup, a.streams)
    all_rec = None
    if (world == 1 and a.config == "c4" and prec != "fp32" and a.scaling == "weak" and not a.no_all_samples and a.streams == 1
            and a.n_res is None and a.total_samples > B):
        # BASELINE configs[3] names 64 samples; the headline line holds one GPU's share of them when they are spread over 8 GPUs.  The same
        # GPU with ALL of them in one batch (what a one-GPU run of the config would do): a few steps of the same schedule, own value and
        # roofline on the same JSON line — the node path then has 8x the rows per launch and stops being latency-bound
        Ba = a.total_samples
        dsa = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": Ba}), diff, dev)
        feats_a, tape_a = sharding.stack_items([sharding.seeded_item(dsa, i, a.seed, diff, T, 0.01) for i in range(Ba)])
        Ka = min(a.all_samples_steps, T)
        el_a, _, _, et_a, clk_a, Bev_a, _ = timed_region(net, prec, Ka, 2, 1, feats=feats_a, tape=tape_a, B=Ba)
        v_a, roof_a = record(prec, Ka, el_a, et_a, clk_a, Bev_a, a.kernel_flags, res_per_step=Ba * N,
                             fwd_flops_per_step=Ba * flops_per_forward(N, inp), B=Ba)
        all_rec = {"dtype": prec, "value": v_a, "unit": "residue*step/s", "samples_per_gpu": Ba, "steps": Ka, "warmup": 2, "ms_per_step": el_a / Ka * 1e3,
                   "roofline": roof_a, "note": f"all {Ba} samples of BASELINE configs[3] batched on ONE GPU, {Ka} consecutive steps of T={T}; same kernels, "
                                               "weights and per-sample seeds as the headline line"}
        del feats_a, tape_a, dsa
        torch.cuda.empty_cache()
    ref_rec = None
    if prec != "fp32" and not a.no_reference_precision:
        # the same workload in the reference's precision (fp32 end to end), a short window of the same schedule: its own value and
        # roofline, so that one command yields both modes (BASELINE.json: the reference never leaves fp32)
        del net
        torch.cuda.empty_cache()
        net32 = ScoreNetwork(conf.model, diff, inpainting=inp, precision="fp32", kernel_flags=a.kernel_flags).load_synthetic(7).to(dev)
        K32 = min(a.reference_steps, T)
        el32, _, _, et32, clk32, Bev32, _ = timed_region(net32, "fp32", K32, 2, 1)
        if rank == 0:
            v32, roof32 = record("fp32", K32, el32, et32, clk32, Bev32, a.kernel_flags)
            ref_rec = {"dtype": "fp32", "value": v32, "unit": "residue*step/s", "steps": K32, "warmup": 2, "ms_per_step": el32 / K32 * 1e3,
                       "precision_mode": PREC_MODE["fp32"], "roofline": roof32,
                       "note": f"same workload and schedule, {K32} consecutive steps of T={T}; per-step backbone RMSD vs the reference ~1e-5 A"}
        del net32

    if rank == 0:
        value, roof = record(prec, K, el, et_ms, clk, B_ev, a.kernel_flags)
        out = {
            "metric": "residue*diffusion-steps/sec", "value": value, "unit": "residue*step/s", "n_gpus": world,
            "n_ranks_seen": n_ranks_seen, "backend": (dist.get_backend() if world > 1 else None),
            "steps": K, "warmup": a.warmup, "ms_per_step": el / K * 1e3, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": f"{a.config}: {'inpainting' if inp else 'de novo'} backbone sampler, "
                                   + (f"{B} different complexes/GPU of N={min(all_len)}..{max(all_len)} (mean {np.mean(all_len):.0f}) padded to {N}, "
                                      if mixed else f"N={N}, {B} samples/GPU ") +
                                   f"batched, {'whole trajectory (priming forward + all steps)' if K == T else f'{K} consecutive steps from the middle of the schedule'}"
                                   f" of T={T}, aux_traj=True, noise_scale 0.1, 17.4M-param synthetic weights, per-sample seeds",
                       "n_res": N, "samples_per_gpu": B, "total_samples": n_total, "num_t": T, "parallelism": f"sample-sharded x{world}, no collective" + (f", {a.streams} sub-batch streams per GPU" if a.streams > 1 else ""),
                       "precision_mode": PREC_MODE[prec], "kernel_flags": a.kernel_flags},
            "roofline": roof,
            # how the timed steps were enqueued: replays of the step graph (the default product path, inference.ReverseLoop) or launch by
            # launch (--eager); host time per step to enqueue them; set-up before the timed region (buffers, noise-tape upload, captures)
            "loop": loop_rec,
        }
        if one_gpu:
            out["one_gpu_test_hook"] = True  # all ranks shared GPU 0 (tests): NOT a multi-GPU measurement
        if d2h is not None:
            out["results_d2h"] = {"seconds": d2h, "bytes": d2h_bytes, "value_including_d2h": res_per_step * K / (el + d2h)}
        if all_rec is not None:
            out["all_samples_one_gpu"] = all_rec
            out.update({"all64_value": all_rec["value"], "all64_ms_per_step": all_rec["ms_per_step"],
                        "all64_roofline_frac": all_rec["roofline"]["frac"], "all64_whole_forward_frac": all_rec["roofline"]["whole_forward_frac"]})
        if ref_rec is not None:
            out["reference_precision"] = ref_rec
            out.update({"fp32_value": ref_rec["value"], "fp32_ms_per_step": ref_rec["ms_per_step"],
                        "fp32_roofline_frac": ref_rec["roofline"]["frac"], "fp32_whole_forward_frac": ref_rec["roofline"]["whole_forward_frac"]})
        out["whole_forward_frac"] = roof["whole_forward_frac"]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(300 if not inp else min(N, 300), config.base_config(), 7)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
