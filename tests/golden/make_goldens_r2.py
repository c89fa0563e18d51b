"""Round-2 golden fixtures: the benchmarked sizes, a 4-chain TCR-pMHC-like complex, a masked res_mask, a bb_gain = 0.3
trajectory, and feature dicts captured from the reference samplers (this container only; imports the reference).

    python tests/golden/make_goldens_r2.py [name ...]      # writes tests/golden/*.npz (all, or the named ones)

Same recipe as make_goldens.py (whose fixtures stay untouched): reference *source* on torch 2.10 / numpy 2.2 / scipy 1.15,
weights regenerated from framedipt_amd.weights.synth_state_dict.  Large fixtures keep the outputs, the node representation
after every block and a few rows of the pair representation (a few MB each).
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_goldens as mg  # noqa: E402  (installs the import stubs, imports the reference)
import refharness as rh  # noqa: E402
import torch  # noqa: E402
from framedipt_amd import weights as W  # noqa: E402
from openfold.utils import rigid_utils as ru  # noqa: E402

np32 = mg.np32


def complex_feats(chain_lens, windows, rng, diff, res_gap=200):
    """Synthetic multi-chain complex (seq_idx gap `res_gap` between chains, framedipt/data/utils.py:745-890 convention) with
    diffused windows [(start, stop), ...] in flattened residue indices."""
    n = int(sum(chain_lens))
    q = mg.rand_quats(rng, n)
    tr = np.cumsum(rng.standard_normal((n, 3)) * 2.2, axis=0).astype(np.float32)
    tr -= tr.mean(0)
    gt = ru.Rigid.from_tensor_7(torch.tensor(np.concatenate([q, tr], -1)))
    dm = np.zeros(n)
    for a, b in windows:
        dm[a:b] = 1
    seq_idx, chain_idx, off = [], [], 0
    for c, L in enumerate(chain_lens):
        seq_idx.append(np.arange(L) + off + c * res_gap)
        chain_idx.append(np.full(L, c))
        off += L
    seq_idx, chain_idx = np.concatenate(seq_idx), np.concatenate(chain_idx).astype(np.int64)
    with torch.no_grad():
        ref = diff.sample_ref(n_samples=n, impute=gt, diffuse_mask=dm, chain_index=chain_idx, as_tensor_7=True)
    tors = rng.standard_normal((1, n, 7, 2))
    tors /= np.linalg.norm(tors, axis=-1, keepdims=True)
    return {
        "res_mask": torch.ones(1, n, dtype=torch.float64),
        "seq_idx": torch.tensor(seq_idx)[None],
        "fixed_mask": torch.tensor(1 - dm)[None],
        "torsion_angles_sin_cos": torch.tensor(tors),
        "sc_ca_t": torch.zeros(1, n, 3, dtype=torch.float64),
        "rigids_t": ref["rigids_t"][None],
        "aatype": torch.tensor(rng.integers(0, 20, size=(1, n))),
        "chain_idx": torch.tensor(chain_idx)[None],
    }


def forward_golden(name, cfg, inpainting, t, feats_fn, sc_scale=1.0, trace_rows=(0, 3), bb_gain=W.BB_GAIN, inner=False,
                   res_mask_zero=()):
    """feats_fn(rng, diff) -> feature dict (without t).  inner: also the per-block IPA / LayerNorm / transformer traces."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    diff, model, shapes = mg.build(cfg, inpainting, bb_gain)
    f = feats_fn(rng, diff)
    n = f["rigids_t"].shape[1]
    if sc_scale:
        f["sc_ca_t"] = torch.tensor(f["rigids_t"][..., 4:].numpy() + rng.standard_normal((1, n, 3)).astype(np.float32) * sc_scale)
    if len(res_mask_zero):
        f["res_mask"] = f["res_mask"].clone()
        f["res_mask"][0, list(res_mask_zero)] = 0
    f["t"] = torch.tensor([t], dtype=torch.float32)
    g = mg.feats_np(f)
    trace, hooks, rows = {}, [], list(trace_rows)
    trunk = model.score_model.trunk

    def hk(key, fn=lambda o: o):
        def _h(_m, _i, o):
            trace[key] = np32(fn(o))
        return _h

    nb = cfg.model.ipa.num_blocks
    for b in range(nb):
        if inner:
            hooks.append(trunk[f"ipa_{b}"].register_forward_hook(hk(f"tr_ipa_{b}")))
            hooks.append(trunk[f"ipa_ln_{b}"].register_forward_hook(hk(f"tr_ipa_ln_{b}")))
            hooks.append(trunk[f"seq_tfmr_{b}"].register_forward_hook(hk(f"tr_tfmr_{b}")))
        hooks.append(trunk[f"node_transition_{b}"].register_forward_hook(hk(f"tr_node_{b}")))
        hooks.append(trunk[f"bb_update_{b}"].register_forward_hook(hk(f"tr_bbupd_{b}")))
        if b < nb - 1:
            hooks.append(trunk[f"edge_transition_{b}"].register_forward_hook(hk(f"tr_edge_{b}", lambda o: o[:, rows])))
    hooks.append(model.embedding_layer.register_forward_hook(
        lambda _m, _i, o: trace.update(tr_node_init=np32(o[0]), tr_edge_init=np32(o[1][:, rows]))))
    with torch.no_grad():
        out = model(f)
    for h in hooks:
        h.remove()
    g.update({"out_" + k: np32(v) for k, v in out.items()})
    g.update(trace)
    g["trace_rows"] = np.array(trace_rows)
    g["weight_seed"] = mg.WEIGHT_SEED
    g["bb_gain"] = bb_gain
    np.savez_compressed(os.path.join(HERE, f"fwd_{name}.npz"), **g)
    print(name, n, {k: float(np.abs(v).max()) for k, v in out.items()}, flush=True)


def traj_golden_gain(name, cfg, n, num_t, bb_gain, noise_scale=0.1, min_t=0.01):
    """make_goldens.traj_golden with a bb_gain other than the default (frames move by O(1) A per block)."""
    saved = W.BB_GAIN
    build0 = mg.build
    mg.build = lambda c, inp=False, g=bb_gain: build0(c, inp, g)
    try:
        mg.traj_golden(name, cfg, n, False, num_t, noise_scale, min_t)
    finally:
        mg.build = build0
    p = os.path.join(HERE, f"traj_{name}.npz")
    g = dict(np.load(p))
    g["bb_gain"] = np.float64(bb_gain)
    np.savez_compressed(p, **g)
    assert saved == W.BB_GAIN


def sampler_goldens():
    """Feature dicts as the reference samplers return them (experiments/sampler.py:69-135,267-354): keys, dtypes, shapes and
    values for a fixed seed.  The conditional sampler reads processed-structure pickles through data_utils.process_csv_row;
    here its per-structure feature dict is handed over directly (the pickle / CSV layer is the 'next' row f2)."""
    from experiments import sampler as rs
    from framedipt.diffusion import se3_diffuser
    cfg = rh.load_cfg()
    g = {}
    diff = se3_diffuser.SE3Diffuser(cfg.diffuser)  # np.random.seed(123)
    ds = rs.UnconditionalSampler(rh.to_attr({"min_length": 20, "max_length": 24, "length_step": 4, "samples_per_length": 2}),
                                 diffuser=diff, device="cpu")
    g["uncond_len"] = np.array([len(ds)])
    for i in (0, 3):
        with torch.no_grad():  # (the reference's inference entry point runs the dataset under no_grad as well)
            length, sample_id, feats = ds[i]
        g[f"uncond_{i}_meta"] = np.array([length, sample_id])
        for k, v in feats.items():
            g[f"uncond_{i}_{k}"] = v.numpy()
            g[f"uncond_{i}_{k}_dtype"] = np.array(str(v.dtype))
    # ConditionalSampler item (sampler.py:267-354) on a processed-structure pickle of the reference's own test complex 1fyt (its
    # processed features are the inputs stored in features.npz); metadata handed over directly (no download / Biopython here)
    import pathlib
    import pickle
    import tempfile
    import pandas as pd
    F = dict(np.load(os.path.join(HERE, "features.npz")))
    cf = {k[len("1fyt_in_"):]: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in F.items() if k.startswith("1fyt_in_")}
    with tempfile.TemporaryDirectory() as td:
        pkl = pathlib.Path(td) / "1fyt.pkl"
        with open(pkl, "wb") as f:
            pickle.dump(cf, f)
        n_mod = int(np.sum(cf["max_modeled_idxs"] - cf["min_modeled_idxs"] + 1))
        cs = rs.ConditionalSampler.__new__(rs.ConditionalSampler)
        cs._data_conf = rh.to_attr({"samples": 2, "seed": 123, "redaction": {"redact_min_len": 8, "redact_max_len": 14}})
        cs.metadata = pd.DataFrame([{"pdb_name": "1fyt-assembly1", "processed_path": str(pkl), "modeled_seq_len": n_mod}])
        np.random.seed(77)
        cs._diffuser, cs.device, cs.diffused_masks, cs.rng = diff, "cpu", {}, np.random.default_rng(123)
        with torch.no_grad():
            name, sample_id, feats = cs[1]
    g["cond_meta"] = np.array([name, str(sample_id), str(n_mod)])
    for k, v in feats.items():
        v = v.numpy()
        g[f"cond_{k}"] = v.astype(np.float32) if (v.dtype == np.float64 and v.size > 5000) else v
        g[f"cond_{k}_dtype"] = np.array(str(v.dtype))
    np.savez_compressed(os.path.join(HERE, "sampler_dicts.npz"), **g)
    print("sampler_dicts cond", name, sample_id, n_mod, int((1 - feats["fixed_mask"]).sum()), flush=True)
    print("sampler_dicts", sorted(k for k in g if k.startswith("uncond_0_") and not k.endswith("_dtype")), flush=True)


def ops_r2_goldens():
    """Per-op vectors of round 2: geomstats-fork SO(3) exp / log / omega (framedipt/diffusion/so3_utils.py), Rigid.from_3_points /
    from_tensor_4x4 (openfold/utils/rigid_utils.py), the unscaled R^3 score, create_redacted_regions / pad_feats
    (framedipt/data/utils.py)."""
    from framedipt.data import utils as du
    from framedipt.diffusion import se3_diffuser, so3_utils
    rng = np.random.default_rng(21)
    g = {}
    n = 48
    rv = rng.standard_normal((n, 3)) * 1.3
    rv[0] = 0
    rv[1] = [1e-6, 0, 0]
    rv[2] = [0, 2e-4, 0]
    ax = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    for i, ang in enumerate([np.pi - 1e-3, np.pi - 5e-3, np.pi - 2e-2, np.pi - 1e-5, 3.0, 0.02]):
        rv[3 + i] = ax * ang
    g["gs_rotvec"] = rv
    R = so3_utils.rot_mat_from_axis_angle_by_exp_map(torch.tensor(rv, dtype=torch.float64))
    g["gs_exp"] = np32(R)            # (float32: skew_symmetric_matrix_from_axis_angle allocates a default-dtype tensor)
    R64 = torch.linalg.matrix_exp(torch.tensor(np.stack([np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]]) for v in rv])))
    g["gs_R64"] = R64.numpy()
    g["gs_omega"] = so3_utils.omega(R64).numpy()
    g["gs_log"] = so3_utils.rotation_vector_from_matrix(R64.float()).numpy()
    g["gs_log_in"] = R64.float().numpy()
    big = rng.standard_normal((12, 3)) * 4.0
    g["gs_reg_in"], g["gs_reg"] = big, so3_utils.regularize(torch.tensor(big, dtype=torch.float32)).numpy()  # (the fork only runs in float32)
    p1, p2, p3 = (rng.standard_normal((n, 3)).astype(np.float32) * 3 for _ in range(3))
    r = ru.Rigid.from_3_points(torch.tensor(p1), torch.tensor(p2), torch.tensor(p3))
    g.update(p3_a=p1, p3_o=p2, p3_c=p3, p3_rot=np32(r.get_rots().get_rot_mats()), p3_trans=np32(r.get_trans()))
    m44 = r.to_tensor_4x4()
    g["t4x4"] = np32(m44)
    g["t4x4_t7"] = np32(ru.Rigid.from_tensor_4x4(m44).to_tensor_7())
    diff = se3_diffuser.SE3Diffuser(rh.load_cfg().diffuser)
    t1, t2 = rng.standard_normal((1, n, 3)).astype(np.float32) * 5, rng.standard_normal((1, n, 3)).astype(np.float32) * 5
    g.update(ts_t1=t1, ts_t2=t2)
    for i, t in enumerate([0.01, 0.5, 1.0]):
        tt = torch.tensor([t], dtype=torch.float32)
        g[f"ts_unscaled_{i}"] = np32(diff.calc_trans_score(torch.tensor(t1), torch.tensor(t2), tt[:, None, None], use_torch=True, scale=False))
    chain_idx = np.concatenate([np.zeros(40), np.ones(25), np.full(9, 2)]).astype(np.int64)
    res_mask = np.ones(74)
    res_mask[:3] = 0
    res_mask[70:] = 0
    g.update(red_chain_idx=chain_idx, red_res_mask=res_mask)
    for seed in (0, 1, 7):
        g[f"red_{seed}"] = du.create_redacted_regions(chain_idx, res_mask, np.random.default_rng(seed), redact_min_len=5, redact_max_len=12)
    g["red_none"] = du.create_redacted_regions(chain_idx, res_mask, np.random.default_rng(0), None, None)
    feats = {"aatype": torch.arange(6), "rigids_0": torch.tensor(rng.standard_normal((6, 7)).astype(np.float32)),
             "atom37_pos": torch.tensor(rng.standard_normal((6, 37, 3))), "t": torch.tensor(1.0)}
    padded = du.pad_feats(feats, 9, use_torch=True)
    for k, v in feats.items():
        g["pad_in_" + k] = v.numpy()
    for k, v in padded.items():
        g["pad_out_" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "ops_r2.npz"), **g)
    print("ops_r2", len(g), "arrays", flush=True)


def writer_goldens():
    """Files the reference's output writers produce from given arrays (framedipt/analysis/utils.py:76-157 -> protein.py:165-281,
    experiments/utils.py:690-749): stored as bytes next to their inputs (data, not source)."""
    import pathlib
    import tempfile

    from experiments import utils as eu
    from framedipt.analysis import utils as au
    rng = np.random.default_rng(3)
    g = {}
    n, T = 23, 3
    pos = np.zeros((T, n, 37, 3), dtype=np.float32)
    pos[:, :, :5] = rng.standard_normal((T, n, 5, 3)).astype(np.float32) * 30  # backbone atoms N, CA, C, CB, O
    pos[0, 2, 3] = 0  # a missing CB (glycine-like): masked by |x| < 1e-7
    pos[:, 5, :5, 0] = [-123.4567, 999.9994, 0.0004, -0.0005, 1234.5]
    aatype = rng.integers(0, 21, n)
    chain_index = np.concatenate([np.full(9, 3), np.full(8, 7), np.full(6, 12)])
    residue_index = np.concatenate([np.arange(9) + 5, np.arange(8) + 220, np.arange(6) + 441])
    dm = np.zeros(n)
    dm[2:6] = 1
    dm[11:14] = 1
    dm[15:16] = 1
    b_factors = np.tile((dm * 100)[:, None], (1, 37))
    g.update(pos=pos, aatype=aatype, chain_index=chain_index, residue_index=residue_index, diffuse_mask=dm)
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        p1 = au.write_prot_to_pdb(pos[0], td / "sample_0", b_factors=b_factors, aatype=aatype, residue_index=residue_index, chain_index=chain_index)
        p2 = au.write_prot_to_pdb(pos, td / "bb_traj_0", b_factors=b_factors, aatype=aatype, residue_index=residue_index, chain_index=chain_index)
        p3 = au.write_prot_to_pdb(pos[1], td / "plain", no_indexing=True)
        p4 = au.write_prot_to_pdb(pos[0], td / "sample_0", b_factors=b_factors)  # second call: index suffix _2
        g["pdb_names"] = np.array([p.name for p in (p1, p2, p3, p4)])
        for k, p in zip(("pdb_sample", "pdb_traj", "pdb_plain", "pdb_second"), (p1, p2, p3, p4)):
            g[k] = np.frombuffer(p.read_bytes(), dtype=np.uint8)
        seq = "".join("ARNDCQEGHILKMFPSTWYVX"[a] for a in aatype)
        eu.save_diffusion_info(td, "1abc", seq, dm, chain_index)
        g["info_seq"] = np.array(seq)
        g["info_csv"] = np.frombuffer((td / "diffusion_info.csv").read_bytes(), dtype=np.uint8)
        ch, st, en = eu.get_diffused_region_per_chain(dm, chain_index)
        g.update(region_chains=np.array(ch), region_starts=np.array(st), region_ends=np.array(en))
    np.savez_compressed(os.path.join(HERE, "writers.npz"), **g)
    print("writers", list(g["pdb_names"]), len(g["pdb_sample"]), "bytes", flush=True)


def confidence_golden(name, cfg, n, inpainting, num_t, min_t=0.01):
    """EigenFold confidence walk (experiments/utils.py:752-869) of the reference with every per-step quantity captured: the noise
    tape, the frames after each forward-noising step, the model's scores and the two log-probabilities."""
    from experiments import utils as exp_utils
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 7)
    diff, model, _ = mg.build(cfg, inpainting)
    np.random.seed(321)
    f = mg.make_feats(n, rng, inpainting, diff)
    g = mg.feats_np(f)
    q = mg.rand_quats(rng, n)[::-1].copy()  # generic rotations first; (an identity rotation makes the reference itself fail:
    q[-9:] = q[:9]                          #  align_rotation_vectors divides by the angle) -> replace the edge-case rows
    q[3] = mg.rand_quats(rng, 9)[5]         # ... but keep one angle ~ pi - 1e-7
    tr = np.cumsum(rng.standard_normal((n, 3)) * 2.2, axis=0).astype(np.float32)
    tr -= tr.mean(0)
    x0 = np.concatenate([q, tr], -1).astype(np.float32)
    dm = 1.0 - f["fixed_mask"][0].numpy()
    tape, steps = [], []
    orig_normal = np.random.normal

    def rec_normal(*a, **k):
        z = orig_normal(*a, **k)
        tape.append(z.copy())
        return z

    o_fwd, o_lpb, o_lpf = diff.forward, diff.log_prob_backward, diff.log_prob_forward

    def rec_fwd(**kw):
        out = o_fwd(**kw)
        steps.append({"t_1": kw["t_1"], "rot": np32(out.get_rots().get_rot_mats()), "trans": np32(out.get_trans())})
        return out

    def rec_lpb(**kw):
        v = o_lpb(**kw)
        steps[-1].update(t=kw["t"], trans_score=np.asarray(kw["trans_score_t"]), rot_score=np.asarray(kw["rot_score_t"]), lp_backward=float(v))
        return v

    def rec_lpf(**kw):
        v = o_lpf(**kw)
        steps[-1]["lp_forward"] = float(v)
        return v

    diff.forward, diff.log_prob_backward, diff.log_prob_forward = rec_fwd, rec_lpb, rec_lpf
    np.random.normal = rec_normal
    try:
        lp, lps = exp_utils.logp_confidence_score(model=model, diffuser=diff, rigids_t=ru.Rigid.from_tensor_7(torch.tensor(x0)),
                                                   sample_feats={k: v.clone() for k, v in f.items()}, diffuse_mask=dm, num_t=num_t,
                                                   min_t=min_t, device="cpu", self_condition=True)
    finally:
        np.random.normal = orig_normal
    g.update(weight_seed=mg.WEIGHT_SEED, bb_gain=W.BB_GAIN, x0=x0, diffuse_mask=dm, num_t=num_t, min_t=min_t, log_prob=float(lp), log_probs=np.array([float(v) for v in lps]),
             noise_tape=np.stack(tape))  # [2 (T-1), N, 3]: R^3 draw then SO(3) draw per step
    for k in ("t_1", "t", "rot", "trans", "trans_score", "rot_score", "lp_backward", "lp_forward"):
        g["step_" + k] = np.stack([np.asarray(s[k]) for s in steps])
    g["score_dtypes"] = np.array([str(steps[0]["trans_score"].dtype), str(steps[0]["rot_score"].dtype)])
    np.savez_compressed(os.path.join(HERE, f"conf_{name}.npz"), **g)
    print(name, "log_prob", lp, "dtypes", g["score_dtypes"], flush=True)


def feature_tables():
    """Residue-constant index tables of the feature builder (framedipt_amd/data/feature_tables.npz), from
    openfold/np/residue_constants.py the way openfold/data/data_transforms.py:572-645,755-800,891-920 derive them."""
    from openfold.np import residue_constants as rc
    from openfold.data import data_transforms as dt
    base = np.full([21, 8, 3], "", dtype=object)
    base[:, 0, :] = ["C", "CA", "N"]
    base[:, 3, :] = ["CA", "C", "O"]
    gmask = np.zeros((21, 8))
    gmask[:, 0] = gmask[:, 3] = 1
    gmask[:20, 4:] = rc.chi_angles_mask
    for r, letter in enumerate(rc.restypes):
        for c in range(4):
            if rc.chi_angles_mask[r][c]:
                base[r, c + 4, :] = rc.chi_angles_atoms[rc.restype_1to3[letter]][c][1:]
    lut = dict(rc.atom_order)
    lut[""] = 0
    a14_37, a37_14, a14_mask = [], [], []
    for rt in rc.restypes:
        names = rc.restype_name_to_atom14_names[rc.restype_1to3[rt]]
        a14_37.append([(rc.atom_order[n] if n else 0) for n in names])
        i14 = {n: i for i, n in enumerate(names)}
        a37_14.append([i14.get(n, 0) for n in rc.atom_types])
        a14_mask.append([1.0 if n else 0.0 for n in names])
    a14_37.append([0] * 14); a37_14.append([0] * 37); a14_mask.append([0.0] * 14)
    a37_mask = np.zeros((21, 37))
    for r, letter in enumerate(rc.restypes):
        for n in rc.residue_atoms[rc.restype_1to3[letter]]:
            a37_mask[r, rc.atom_order[n]] = 1
    out = {
        "atom_types": np.array(rc.atom_types), "restype_3": np.array([rc.restype_1to3[r] for r in rc.restypes]),
        "rigidgroup_base_atom37_idx": np.vectorize(lambda x: lut[x])(base).astype(np.int64), "rigidgroup_mask": gmask,
        "atom14_to_atom37": np.array(a14_37, dtype=np.int64), "atom37_to_atom14": np.array(a37_14, dtype=np.int64),
        "atom14_mask": np.array(a14_mask), "atom37_mask": a37_mask,
        "chi_atom_indices": np.array(dt.get_chi_atom_indices(), dtype=np.int64),
        "chi_angles_mask": np.array(list(rc.chi_angles_mask) + [[0.0] * 4]), "chi_pi_periodic": np.array(rc.chi_pi_periodic),
    }
    np.savez_compressed(os.path.join(mg.ROOT, "framedipt_amd", "data", "feature_tables.npz"), **out)
    print("feature_tables", {k: v.shape for k, v in out.items()}, flush=True)


def feature_builder_golden():
    """process_csv_row of the reference (framedipt/data/utils.py:745-890) on the three complexes of the reference's own test data
    (tests/data/inference_data/structures/cifs): the processed-structure dict is read from the mmCIF by framedipt_amd.data.mmcif
    (Biopython is not installed here) and stored with the outputs.  One all-chains row per structure + one single-chain row with a cut."""
    import pathlib
    import pickle
    import tempfile
    from framedipt.data import utils as du
    from framedipt_amd.data import mmcif
    cif_dir = pathlib.Path("/root/reference/tests/data/inference_data/structures/cifs")
    g = {}
    for cif in sorted(cif_dir.glob("*.cif")):
        name = cif.stem[:4]
        _, _, _, cf = mmcif.extract_features_from_mmcif(cif)
        for k, v in cf.items():
            g[f"{name}_in_{k}"] = v.astype(np.float32) if k in ("atom_positions", "b_factors", "bb_positions") else v
        cf = {k: (g[f"{name}_in_{k}"].astype(np.float64) if v.dtype.kind == "f" else v) for k, v in cf.items()}
        with tempfile.TemporaryDirectory() as td:
            p = pathlib.Path(td) / f"{name}.pkl"
            with open(p, "wb") as f:
                pickle.dump(cf, f)
            out = du.process_csv_row(p, False, False, None, None)
            rng = np.random.default_rng(11)
            one = du.process_csv_row(p, False, True, rng, 150)
        for tag, o in (("all", out), ("one", one)):
            for k, v in o.items():
                v = v.numpy() if torch.is_tensor(v) else np.asarray(v)
                g[f"{name}_{tag}_{k}"] = v.astype(np.float32) if (v.dtype == np.float64 and k in ("atom37_pos", "atom14_pos")) else v
                g[f"{name}_{tag}_{k}_dtype"] = np.array(str(v.dtype))
        print(name, "N", len(out["aatype"]), "chains", len(np.unique(out["chain_idx"])), "single-chain N", len(one["aatype"]), flush=True)
    np.savez_compressed(os.path.join(HERE, "features.npz"), **g)


def denovo(n):
    return lambda rng, diff: mg.make_feats(n, rng, False, diff)


JOBS = {
    "full_denovo_n128": lambda: forward_golden("full_denovo_n128", rh.load_cfg(), False, 0.5, denovo(128), trace_rows=(0, 3, 64, 127)),
    "full_denovo_n300_t50": lambda: forward_golden("full_denovo_n300_t50", rh.load_cfg(), False, 0.5, denovo(300),
                                                   trace_rows=(0, 3, 150, 299)),
    "full_denovo_n300_t02": lambda: forward_golden("full_denovo_n300_t02", rh.load_cfg(), False, 0.02, denovo(300),
                                                   trace_rows=(0, 3, 150, 299)),
    "full_denovo_n64_inner": lambda: forward_golden("full_denovo_n64_inner", rh.load_cfg(), False, 0.3, denovo(64), inner=True),
    "full_denovo_n64_masked": lambda: forward_golden("full_denovo_n64_masked", rh.load_cfg(), False, 0.5, denovo(64),
                                                     res_mask_zero=(5, 6, 31, 61, 62, 63)),
    "full_inpaint_n724_4chain": lambda: forward_golden(
        "full_inpaint_n724_4chain", rh.load_cfg(inpainting=True), True, 0.4,
        lambda rng, diff: complex_feats((200, 240, 9, 275), ((92, 106), (330, 345)), rng, diff), trace_rows=(0, 100, 444, 723)),
    "full_inpaint_n1000": lambda: forward_golden(
        "full_inpaint_n1000", rh.load_cfg(inpainting=True), True, 0.6,
        lambda rng, diff: complex_feats((500, 500), ((40, 90),), rng, diff), trace_rows=(0, 60, 999)),
    "traj_full_denovo_n300_T5": lambda: mg.traj_golden("full_denovo_n300_T5", rh.load_cfg(), 300, False, 5),
    "traj_full_denovo_n64_T20_gain03": lambda: traj_golden_gain("full_denovo_n64_T20_gain03", rh.load_cfg(), 64, 20, 0.3),
    "conf_small_denovo_n24_T6": lambda: confidence_golden("small_denovo_n24_T6", rh.small_model_cfg(rh.load_cfg()), 24, False, 6),
    "conf_full_inpaint_n40_T5": lambda: confidence_golden("full_inpaint_n40_T5", rh.load_cfg(inpainting=True), 40, True, 5),
    "feature_tables": feature_tables,
    "features": feature_builder_golden,
    "sampler_dicts": sampler_goldens,
    "ops_r2": ops_r2_goldens,
    "writers": writer_goldens,
}


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(JOBS)):
        JOBS[name]()
