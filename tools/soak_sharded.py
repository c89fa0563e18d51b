"""Soak of the two shipped multi-queue paths (round-3 review, item 2): `run_sharded` with its D2H copy stream under compute, and two ranks
on ONE GPU (FDIPT_ONE_GPU=1), at TCR-pMHC sizes (the three ~810-residue complexes of tests/golden/features.npz), against the one-rank run:
every output file must be byte-identical in every repetition.

    python tools/soak_sharded.py REPS [num_t] [samples_per_structure]
"""
import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
NUM_T = sys.argv[2] if len(sys.argv) > 2 else "8"
SAMPLES = sys.argv[3] if len(sys.argv) > 3 else "3"


def tree(out_dir):
    files = sorted(os.path.relpath(os.path.join(d, f), out_dir) for d, _, fs in os.walk(out_dir) for f in fs if f.endswith((".pdb", ".csv")))
    return {f: open(os.path.join(out_dir, f), "rb").read() for f in files}


def run(data, out_dir, world, port):
    env = dict(os.environ, FDIPT_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "framedipt_amd.run_sharded", "--out-dir", out_dir, "--download-dir", data,
           "--samples-per-structure", SAMPLES, "--num-t", NUM_T, "--max-batch", "3", "--precision", "fp16", "--full-trajectory"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stderr[-2000:]
    return tree(out_dir)


with tempfile.TemporaryDirectory() as td:
    F = dict(np.load(os.path.join(ROOT, "tests", "golden", "features.npz")))
    data = os.path.join(td, "data")
    os.makedirs(os.path.join(data, "processed"))
    rows = []
    for name in ("1fyt", "5ksa", "7t2d"):
        cf = {k[len(name) + 4:]: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in F.items() if k.startswith(name + "_in_")}
        with open(os.path.join(data, "processed", f"{name}.pkl"), "wb") as f:
            pickle.dump(cf, f)
        rows.append({"pdb_name": f"{name}-assembly1", "processed_path": os.path.join(data, "processed", f"{name}.pkl"),
                     "modeled_seq_len": int(np.sum(np.asarray(cf["max_modeled_idxs"]) - np.asarray(cf["min_modeled_idxs"]) + 1))})
    pd.DataFrame(rows).to_csv(os.path.join(data, "processed", "metadata.csv"), index=False)
    ref = run(data, os.path.join(td, "ref"), 1, 29700)
    print(f"reference run: {len(ref)} files, {sum(len(v) for v in ref.values()) / 1e6:.1f} MB", flush=True)
    bad = {1: 0, 2: 0}
    for rep in range(REPS):
        for world in (1, 2):
            got = run(data, os.path.join(td, f"w{world}_{rep}"), world, 29701 + 2 * rep + world)
            diff = [f for f in ref if got.get(f) != ref[f]] + [f for f in got if f not in ref]
            if diff:
                bad[world] += 1
                print(f"rep {rep} world {world}: {len(diff)} files differ, e.g. {diff[:3]}", flush=True)
    print(f"soak_sharded: {REPS} repetitions, num_t {NUM_T}, {SAMPLES} samples per structure (N = 810 / 820 / 801, batches of 3, full trajectories through the copy "
          f"stream): runs with a differing file: one rank {bad[1]}, two ranks on one GPU {bad[2]}")
