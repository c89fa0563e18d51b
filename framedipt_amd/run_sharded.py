"""Sample-sharded sampling over the GPUs of one node (torchrun entry; SURVEY.md section 8e, experiments/inference.py:198-242 de novo,
:244-389 inpainting).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        -m framedipt_amd.run_sharded --out-dir samples/ --min-length 300 --max-length 300 --samples-per-length 64 --num-t 500

Inpainting (BASELINE configs[2]: every structure of a database CSV, k samples each): ``--download-dir D`` names the reference's
``data.download_dir`` (``D/cifs/*.cif`` and / or ``D/processed/metadata.csv``), ``--csv`` its ``data.data_path``, ``--tcr`` selects
``TCRSampler`` (CDR masks must be in the processed features), ``--samples-per-structure k``; the outputs are then written in the
reference's directory layout (``<out>/<pdb>_length_<L>/{<pdb>_1.pdb, diffusion_info.csv, sample_<i>/sample_<i>_1.pdb [, bb_traj_<i>_1.pdb,
x0_traj_<i>_1.pdb]}``, experiments/inference.py:267-389,480-556) through ``framedipt_amd.output``:

    python -m torch.distributed.run ... -m framedipt_amd.run_sharded --out-dir samples/ --download-dir data/ --csv database/TCR_pMHC_I.csv \
        --tcr --samples-per-structure 5 --num-t 100

One process per GPU.  Dataset items (length, sample index) are dealt round-robin to the ranks; every rank runs its items -
batched by equal length - through ``inference_fn`` and writes one ``sample_<item>.npz`` per item into the shared output
directory; rank 0 writes ``manifest.json`` once every rank is done.  No collective touches the data path: the only
``torch.distributed`` call is the final barrier.  Per-sample seeds (``seed + item``) make a sample's trajectory independent
of the world size (framedipt_amd/sharding.py).
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np


def run_rank(dataset, diffuser, run_batch, rank: int, world: int, out_dir: str, seed: int, num_t: int, min_t: float,
             max_batch: int = 8, keep=("prot_traj",), final_only: bool = True, mixed: bool = True, write_item=None):
    """Run this rank's share of ``dataset``.  ``run_batch(feats, tape) -> dict of arrays with a batch axis at dim 1`` (the
    keys of ``inference_fn``).  ``mixed``: samples of similar (not only equal) length share a batch, padded with res_mask = 0
    rows (sharding.batches_mixed / stack_items_padded; a batch never spans two kernel-selection classes, so a sample's bits do not
    depend on its batch mates); results are cut back to each sample's own length.
    Returns the list of records written by this rank."""
    from . import sharding
    os.makedirs(out_dir, exist_ok=True)
    mine = sharding.shard_indices(len(dataset), rank, world)
    items = [sharding.seeded_item(dataset, i, seed, diffuser, num_t, min_t) for i in mine]
    lengths = [int(it[2]["rigids_t"].shape[1]) for it in items]
    records = []
    groups = sharding.batches_mixed(lengths, max_batch) if mixed else sharding.batches_by_length(lengths, max_batch)

    def write(group, res):
        for b, p in enumerate(group):
            item, (name, sample_i) = mine[p], items[p][:2]
            n = lengths[p]  # (every per-residue output carries the residues on the axis behind the batch)
            arrays = {k: (np.asarray(res[k])[0, b, :n] if final_only else np.asarray(res[k])[:, b, :n]) for k in keep}
            if write_item is not None:  # caller-defined layout (the reference's directories for inpainting runs)
                path = write_item(int(item), name, int(sample_i), arrays, items[p][2])
            else:
                path = os.path.join(out_dir, f"sample_{item:06d}.npz")
                np.savez(path, item=item, name=str(name), sample_i=int(sample_i), **arrays)
            records.append({"item": int(item), "name": str(name), "sample_i": int(sample_i), "n_res": lengths[p],
                            "rank": rank, "file": os.path.relpath(str(path), out_dir)})

    # The trajectories of batch k leave the device (pinned buffers, a copy stream) and are written to disk while batch k + 1
    # computes: the loop enqueues k + 1 before it waits for k's copy.  (At 8 samples x T = 500 x N = 300 the copy is 1.1 GB =
    # 8 - 12 % of a batch's wall time when it is not overlapped.)  run_batch may also return NumPy arrays (no overlap then).
    pending = None  # (group, {key: pinned host tensor}, copy-done event)
    for group in groups:
        if mixed:
            feats, tape, _ = sharding.stack_items_padded([items[p] for p in group])
        else:
            feats, tape = sharding.stack_items([items[p] for p in group])
        res = run_batch(feats, tape)
        on_device = all(hasattr(res[k], "is_cuda") and res[k].is_cuda for k in keep)
        if not on_device:
            write(group, res)
            continue
        import torch
        dev = res[keep[0]].device
        with torch.cuda.device(dev):
            done = torch.cuda.Event()
            done.record()                       # batch k's kernels are enqueued up to here
            if pending is not None:             # batch k - 1: its copy ran under batch k - 1's tail / this batch's set-up
                pending[2].synchronize()
                write(pending[0], {k: v.numpy() for k, v in pending[1].items()})
            copy_stream = _copy_stream(dev)
            copy_stream.wait_event(done)
            host = {}
            with torch.cuda.stream(copy_stream):
                for k in keep:
                    src = res[k][:1] if final_only else res[k]   # (only what is written crosses PCIe)
                    src = src.contiguous()
                    src.record_stream(copy_stream)
                    host[k] = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                    host[k].copy_(src, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(copy_stream)
            pending = (group, host, copied)
    if pending is not None:
        pending[2].synchronize()
        write(pending[0], {k: v.numpy() for k, v in pending[1].items()})
    with open(os.path.join(out_dir, f"records_rank{rank}.json"), "w") as f:
        json.dump(records, f)
    return records


_COPY_STREAMS: dict = {}


def _copy_stream(dev):
    import torch
    key = str(dev)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


def write_manifest(out_dir: str, world: int, n_items: int, meta: dict):
    """Rank 0, after the barrier: merge the per-rank record files; raises if an item is missing."""
    recs = []
    for r in range(world):
        with open(os.path.join(out_dir, f"records_rank{r}.json")) as f:
            recs += json.load(f)
    recs.sort(key=lambda x: x["item"])
    if [x["item"] for x in recs] != list(range(n_items)):
        raise RuntimeError("sharded run is incomplete: items " + str(sorted(set(range(n_items)) - {x["item"] for x in recs})))
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump({"n_items": n_items, "world_size": world, **meta, "samples": recs}, f, indent=1)
    return recs


def reference_layout_writer(out_dir: str, net, final_only: bool):
    """``write_item`` for inpainting runs: what ``Inference.run_conditional_sampling`` leaves on disk per sample
    (experiments/inference.py:250-389): ``<pdb>_length_<L>/`` with the ground-truth structure ``<pdb>_1.pdb`` (b-factor 100 = diffused) and
    ``diffusion_info.csv`` — written once per structure, by whichever rank gets there first: the content does not depend on the rank, the
    file appears atomically — and ``sample_<i>/sample_<i>_1.pdb`` (+ ``bb_traj_<i>_1.pdb`` / ``x0_traj_<i>_1.pdb`` with the trajectories kept)."""
    import pathlib
    import tempfile

    from . import inference, output

    def once(path: pathlib.Path, make):
        if path.exists():
            return
        tmp = pathlib.Path(tempfile.mkdtemp(dir=path.parent, prefix=".tmp_"))
        try:
            made = make(tmp)
            os.replace(made, path)
        finally:
            for f in tmp.iterdir():
                f.unlink()
            tmp.rmdir()

    def write_item(item, name, sample_i, arrays, feats):
        host = lambda k: feats[k][0].detach().cpu().numpy()  # noqa: E731
        res_mask, fixed = host("res_mask").astype(bool), host("fixed_mask").astype(bool)
        diffused = (1 - fixed) * res_mask
        aatype, residue_index, chain_index = host("aatype"), host("residue_index"), host("chain_idx")
        length_dir = pathlib.Path(out_dir) / f"{name}_length_{int(res_mask.sum() - (fixed * res_mask).sum())}"
        length_dir.mkdir(parents=True, exist_ok=True)
        kw = dict(aatype=aatype[res_mask], residue_index=residue_index[res_mask], chain_index=chain_index[res_mask])

        def gt(tmp):
            pos = inference.get_atom_positions_from_rigids(net, feats["rigids_0"], feats["torsion_angles_sin_cos"][..., 2, :], feats["aatype"])[0]
            b_factors = np.tile((diffused.astype(bool) * 100)[:, None], (1, 37))
            return output.write_prot_to_pdb(pos[res_mask], tmp / str(name), b_factors=b_factors[res_mask], overwrite=True, **kw)

        once(length_dir / f"{name}_1.pdb", gt)

        def info(tmp):
            output.save_diffusion_info(tmp, str(name), output.aatype_to_seq(aatype[res_mask]), diffused[res_mask], chain_index[res_mask])
            return tmp / "diffusion_info.csv"

        once(length_dir / "diffusion_info.csv", info)
        sample_dir = length_dir / f"sample_{sample_i}"
        sample_dir.mkdir(parents=True, exist_ok=True)
        for old in sample_dir.glob("*.pdb"):  # (a re-run replaces the sample: write_prot_to_pdb would otherwise number a second file)
            old.unlink()
        prot, x0 = arrays["prot_traj"], arrays.get("rigid_0_traj")
        n = prot.shape[-3]
        if final_only:  # [N,37,3]: the sample only
            prot, x0 = prot[None], None
        paths = output.save_traj(prot[:, res_mask[:n]], x0[:, res_mask[:n]] if x0 is not None else prot[:, res_mask[:n]],
                                 diffused[res_mask], sample_dir, sample_i, save_backbone_trajectory=not final_only,
                                 save_pred_x0_trajectory=not final_only and x0 is not None, **kw)
        return paths["sample_path"]

    return write_item


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--min-length", type=int, default=300)
    ap.add_argument("--max-length", type=int, default=300)
    ap.add_argument("--length-step", type=int, default=1)
    ap.add_argument("--samples-per-length", type=int, default=8)
    ap.add_argument("--num-t", type=int, default=500)
    ap.add_argument("--min-t", type=float, default=0.01)
    ap.add_argument("--noise-scale", type=float, default=0.1)
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--weights-seed", type=int, default=7, help="synthetic weights (no checkpoint is available offline)")
    ap.add_argument("--full-trajectory", action="store_true", help="store every step of prot_traj instead of the final sample")
    # inpainting runs (experiments/inference.py:244-389)
    ap.add_argument("--download-dir", default=None, help="inpainting: the reference's data.download_dir (cifs/ and / or processed/metadata.csv)")
    ap.add_argument("--csv", default=None, help="inpainting: the reference's data.data_path (database CSV with a pdb_id column; TCR chain columns with --tcr)")
    ap.add_argument("--tcr", action="store_true", help="TCRSampler: CDR loops are redesigned (masks from the processed features)")
    ap.add_argument("--samples-per-structure", type=int, default=5)
    ap.add_argument("--redact-min-len", type=int, default=8)
    ap.add_argument("--redact-max-len", type=int, default=14)
    ap.add_argument("--no-input-aatype", action="store_true", help="inference.input_aatype = False (default True: backbone atoms are built with the true residue types)")
    ap.add_argument("--allow-shared-gpu", action="store_true", help="run although another compute process holds queues on this rank's GPU "
                    "(refused by default: kernels of two processes on one GPU can corrupt each other's results, DESIGN.md section 6)")
    ap.add_argument("--verify", type=int, default=0, help="inference_fn(verify=k): the forward of every k-th step runs twice and must reproduce its bits")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    from . import config, inference
    from .diffusion import SE3Diffuser
    from .model import ScoreNetwork
    from .sampler import UnconditionalSampler

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    # FDIPT_ONE_GPU=1 (tests only): all ranks share GPU 0 and rendezvous over gloo — the multi-rank path on a one-GPU box, where
    # RCCL refuses two ranks on one device
    one_gpu = os.environ.get("FDIPT_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    # one process per GPU is the contract: refuse a GPU that another compute process is using (the one-GPU test hook shares it on
    # purpose and serialises its ranks with a file lock below).  BEFORE the rendezvous: a refusing rank then never joins the group and
    # the launcher tears the job down, instead of the other ranks waiting in the final barrier for its time-out (round-5 advisor)
    from . import gpu_guard
    if one_gpu or a.allow_shared_gpu:
        os.environ["FDIPT_SHARED_GPU"] = "allow"
    else:
        gpu_guard.check(dev, policy=os.environ.get("FDIPT_SHARED_GPU") or "refuse", what="run_sharded")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_gpu else "nccl", rank=rank, world_size=world)
    inp = a.download_dir is not None
    conf = config.base_config(inpainting=inp)
    diff = SE3Diffuser(conf.diffuser, device=dev)
    net = ScoreNetwork(conf.model, diff, inpainting=inp, precision=a.precision).load_synthetic(a.weights_seed).to(dev)
    write_item, keep = None, ("prot_traj",)
    if inp:
        from .sampler import ConditionalSampler, TCRSampler
        data_conf = config.to_conf({"download_dir": a.download_dir, "data_path": a.csv, "samples": a.samples_per_structure, "seed": a.seed,
                                    "redaction": {"redact_min_len": a.redact_min_len, "redact_max_len": a.redact_max_len},
                                    "cdr_loops": ["CDR3"], "first_assembly": True})
        ds = (TCRSampler if a.tcr else ConditionalSampler)(data_conf, diff, dev)
        write_item = reference_layout_writer(a.out_dir, net, final_only=not a.full_trajectory)
        keep = ("prot_traj",) if not a.full_trajectory else ("prot_traj", "rigid_0_traj")
    else:
        ds = UnconditionalSampler(config.to_conf({"min_length": a.min_length, "max_length": a.max_length,
                                                  "length_step": a.length_step, "samples_per_length": a.samples_per_length}), diff, dev)

    def run_batch(feats, tape):
        def go():
            return inference.inference_fn(net, diff, feats, num_t=a.num_t, min_t=a.min_t, aux_traj=True, noise_scale=a.noise_scale,
                                          noise_tape=tape, return_device=True, inpainting=inp, verify=a.verify,
                                          input_aatype=inp and not a.no_input_aatype)  # (run_rank overlaps the D2H copy with the next batch)
        if not (one_gpu and world > 1):
            return go()
        # FDIPT_ONE_GPU (tests): the ranks share one GPU.  Kernels of two processes must not be co-resident on it (DESIGN.md section 6:
        # a half-precision MFMA kernel next to another kernel's waves corrupts single residues — 1 of 12 two-rank soak runs at N = 810
        # differed in one sample), so the ranks take turns on the device: a file lock around each batch, released once its kernels are done.
        # Production runs are one process per GPU and never take this path.
        return one_gpu_turn(go)

    def one_gpu_turn(fn):
        import fcntl
        os.makedirs(a.out_dir, exist_ok=True)
        with open(os.path.join(a.out_dir, ".one_gpu.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                res = fn()
                torch.cuda.synchronize()
                return res
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)

    if write_item is not None and one_gpu and world > 1:
        # the writer's ground-truth backbone launch (reference_layout_writer -> get_atom_positions_from_rigids) is GPU work as well
        plain_write = write_item
        write_item = lambda *args, **kw: one_gpu_turn(lambda: plain_write(*args, **kw))  # noqa: E731

    t0 = time.perf_counter()
    recs = run_rank(ds, diff, run_batch, rank, world, a.out_dir, a.seed, a.num_t, a.min_t, a.max_batch, keep=keep,
                    final_only=not a.full_trajectory, write_item=write_item)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if rank == 0:
        el = time.perf_counter() - t0
        allrecs = write_manifest(a.out_dir, world, len(ds), {"num_t": a.num_t, "precision": a.precision, "seed": a.seed, "wall_s": el})
        print(f"{sum(r['n_res'] for r in allrecs) * a.num_t / el:.0f} residue*steps/s (incl. model set-up and file output)")
        print(f"{len(ds)} samples on {world} GPU(s) in {el:.1f} s -> {a.out_dir}/manifest.json", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
