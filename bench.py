#!/usr/bin/env python
"""Headline benchmark: residue x diffusion-steps / s of the de novo sampler (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], weak scaling): de novo backbone, length 300, 8 samples per GPU batched in one
trajectory batch, 500-step schedule, synthetic weights of the full 17.4 M-parameter network, bf16 GEMM operands.
A "step" = one reverse-diffusion step of the whole batch (score-network forward + fused SE(3) reverse step +
backbone atoms); K consecutive steps of the 500-step schedule are timed after W warm-up steps, inputs and the
noise tape already resident in HBM.  value = B*N*K*n_gpus / max-over-ranks(time).

Also reported on the same JSON line:
  roofline     : the dominant kernel (EdgeTransition, 89 % of reference FLOPs) timed with HIP events on the
                 launch stream inside the timed region; achieved = reference-formulation FLOPs per launch / duration.
  cpu_baseline : the NumPy oracle (port of the reference loop) timed on this box's host cores on a bounded sample.
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

ET_FLOPS_PER_PAIR = 688128.0  # 2*(2*384^2 + 384*128): EdgeTransition, reference formulation
PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md


def pmc_traffic(precision: str, n: int, b: int):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (profiles/), only when the
    bench runs the profiled configuration; the counters cannot be read from inside this process."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_edge_transition.json")) as f:
            rec = json.load(f)
        w = rec["workload"]
        if (w["precision"], w["n_res"], w["samples_per_gpu"]) == (precision, n, b):
            return rec["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def flops_per_forward(n: int) -> float:
    """SURVEY.md 8d / BASELINE.md section 3 (de novo)."""
    return 2248960.0 * n * n + 32.4e6 * n


def make_batch(sampler, B):
    import torch
    items = [sampler[i][2] for i in range(B)]
    return {k: torch.cat([it[k] for it in items], dim=0) for k in items[0]}


def cpu_baseline(n: int, conf, seed: int, steps: int = 2):
    """NumPy oracle, same loop (x_T + priming + `steps` reverse steps), all host cores via BLAS threads."""
    from framedipt_amd import weights as W
    from oracle import diffuser as od
    from oracle import inference as oi
    from oracle.score_network import ScoreNetwork as OracleNet
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    odiff = od.SE3Diffuser(conf.diffuser)
    net = OracleNet(conf.model, odiff, W.synth_state_dict(W.param_shapes(conf.model), seed), tables=tables)
    feats = oi.unconditional_feats(odiff, n)
    # steps of a 500-step schedule: emulate with num_t=500 but run only the first `steps`
    tp = np.ones((1,), dtype=np.float32)
    sched = np.linspace(0.01, 1.0, 500)[::-1]
    t0 = time.perf_counter()
    feats = oi.set_t_feats(feats, sched[0], tp, odiff)
    feats["sc_ca_t"] = net(feats)["rigids"][..., 4:]
    for k in range(steps):
        feats, *_ = oi.one_step(net, odiff, feats, sched[k], 0.01, 1 / 500, tp, noise_scale=0.1)
    el = time.perf_counter() - t0
    fwd = steps + 1
    return {"value": n * steps / (el * steps / fwd), "unit": "residue*step/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"NumPy oracle, de novo N={n}, B=1, {fwd} forwards + {steps} reverse steps of the T=500 schedule "
                      f"({el:.1f} s wall, priming forward amortised as (T+1)/T)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-res", type=int, default=300)
    ap.add_argument("--samples-per-gpu", type=int, default=8)
    ap.add_argument("--num-t", type=int, default=500)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=123)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from framedipt_amd import _lib, config, inference, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler

    lib = _lib.load()
    conf = config.base_config()
    N, B, T = a.n_res, a.samples_per_gpu, a.num_t
    if a.warmup + a.steps > T:
        raise SystemExit("warmup + steps exceeds the schedule length")
    # independent samples: global sample index -> rank (round-robin), per-sample seed = seed + index
    my_samples = sharding.shard_indices(B * world, rank, world)
    conf.diffuser.so3.seed = conf.diffuser.r3.seed = a.seed + rank
    diff = SE3Diffuser(conf.diffuser, device=dev)
    net = ScoreNetwork(conf.model, diff, precision=a.precision).load_synthetic(7).to(dev)
    sampler = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1,
                                                   "samples_per_length": len(my_samples)}), diff, dev)
    feats = make_batch(sampler, len(my_samples))
    loop = inference.ReverseLoop(net, diff, feats, num_t=T, min_t=0.01, aux_traj=False, noise_scale=0.1)
    st = loop.st
    nb = conf.model.ipa.num_blocks
    # HIP events around every EdgeTransition launch (recorded on the launch stream by the library)
    n_ev = nb - 1
    ev_s, ev_e = (C.c_void_p * nb)(), (C.c_void_p * nb)()
    for i in range(n_ev):
        for arr in (ev_s, ev_e):
            h = C.c_void_p()
            _lib.check(lib.fdipt_event_create(C.byref(h)))
            arr[i] = h

    loop.prime()
    for k in range(a.warmup):
        loop.step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    et_ms, t0 = [], time.perf_counter()
    for k in range(a.warmup, a.warmup + a.steps):
        # the EdgeTransition launches of every 8th step are bracketed by HIP events on the launch stream (an event record costs
        # ~6 us of queue time on either side of a launch, and reading it back syncs the stream: sampled, not every step)
        sample = (k - a.warmup) % 8 == 0
        st.ev_start, st.ev_stop = (ev_s, ev_e) if sample else (None, None)
        loop.step(k)
        if sample:
            for i in range(n_ev):
                ms = C.c_float()
                _lib.check(lib.fdipt_event_elapsed_ms(ev_s[i], ev_e[i], C.byref(ms)))
                et_ms.append(ms.value)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st.ev_start = st.ev_stop = None
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        dist.barrier()

    if rank == 0:
        res_steps = B * N * a.steps * world
        value = res_steps / el
        et = float(np.mean(et_ms)) * 1e-3
        et_flops = ET_FLOPS_PER_PAIR * B * N * N
        peak = PEAK_TFLOPS[a.precision]
        achieved = et_flops / et / 1e12
        fwd_tflops = value / world * (flops_per_forward(N) / N) / 1e12  # whole-forward view, per GPU
        out = {
            "metric": "residue*diffusion-steps/sec", "value": value, "unit": "residue*step/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"de novo backbone sampler, N={N}, {B} samples/GPU batched, {a.steps}-step window of "
                                   f"the T={T} schedule, noise_scale 0.1, 17.4M-param synthetic weights",
                       "n_res": N, "samples_per_gpu": B, "num_t": T, "parallelism": f"sample-sharded x{world}, no collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic(a.precision, N, B), "kernel": "edge_transition4_kernel" if a.precision == "fp16" else "edge_transition_kernel",
                         "avg_launch_ms": et * 1e3, "flops_per_launch": et_flops,
                         "whole_forward_tflops": fwd_tflops, "whole_forward_frac": fwd_tflops / peak},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, conf, 7)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
