"""Score network with the call surface of ``framedipt.model.score_network.ScoreNetwork``.

``model(feats) -> {"psi","rot_score","trans_score","rigids","atom37","atom14"}`` for a feature dict with a leading
batch dimension (``framedipt/model/score_network.py:218-275``).  The whole forward is one C-ABI call
(``fdipt_score_forward``) that enqueues hand-written HIP kernels; this class only owns device buffers and the
host-side per-call scalars (timestep embedding, IGSO(3) sigma).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from .. import _lib, embedding, residue_tables, weights


PRECISIONS = {"fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "fp16": _lib.PREC_F16, "f16": _lib.PREC_F16,
              "bf16": _lib.PREC_BF16}  # bf16: only with a library built with -DFDIPT_HALF_BF16


def dims_from_conf(model_conf, diffuser_conf, inpainting: bool, precision: int, kernel_flags: int = 0) -> _lib.Dims:
    i, e = model_conf.ipa, model_conf.embed
    return _lib.Dims(i.c_s, i.c_z, i.c_hidden, i.c_skip, i.no_heads, i.no_qk_points, i.no_v_points,
                     i.seq_tfmr_num_heads, i.seq_tfmr_num_layers, i.num_blocks, e.index_embed_size, e.num_bins,
                     int(bool(inpainting or model_conf.input_aatype)), precision, kernel_flags, float(e.min_bin), float(e.max_bin),
                     float(i.coordinate_scaling), float(diffuser_conf.r3.min_b), float(diffuser_conf.r3.max_b))


def preprocess_aatype(aatype, fixed_mask, inpainting: bool, input_aatype: bool):
    """framedipt/data/utils.py:565-610 (index bookkeeping on the feature dict)."""
    if aatype is None or (not inpainting and not input_aatype):
        return None
    aatype = aatype.to(torch.int64)
    if not input_aatype:
        aatype = torch.where(fixed_mask.bool(), aatype, torch.full_like(aatype, 20))
    return aatype


class BatchState:
    """Device buffers of one batch of B equally sized samples (constant along a trajectory)."""

    def __init__(self, net: "ScoreNetwork", seq_idx: torch.Tensor, trace: bool = False, trace_inner: bool = False):
        lib = _lib.load()
        self.net = net
        dev = net.device
        B, N = seq_idx.shape
        self.B, self.N = B, N
        d = net.dims
        sidx = seq_idx.detach().cpu().numpy().astype(np.int64)
        self.seq_idx = torch.as_tensor(sidx.astype(np.int32), device=dev)
        E = d.index_embed
        self.idx_emb = torch.as_tensor(embedding.get_index_embedding(sidx, E), device=dev)
        rng = int(sidx.max() - sidx.min())
        self.rel_off, self.n_rel = rng, 2 * rng + 1
        rel = np.arange(self.n_rel) - rng
        rel_emb = np.broadcast_to(embedding.get_index_embedding(rel, E)[None], (B, self.n_rel, E)).copy()
        rel_emb = torch.as_tensor(rel_emb, device=dev)
        nb = lib.fdipt_setup_bytes(C.byref(d), B, N, self.n_rel)
        self.setup = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.fdipt_sample_setup(C.byref(d), _lib.ptr(net.params), _lib.ptr(net.derived), B, N, self.n_rel,
                                          _lib.ptr(rel_emb), _lib.ptr(self.setup), _lib.stream_ptr()), "sample_setup")
        self.ws_bytes = lib.fdipt_forward_workspace_bytes(C.byref(d), B, N)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.psi = torch.empty(B, N, 2, **f32)
        self.rot_score = torch.empty(B, N, 3, dtype=torch.float64, device=dev)
        self.trans_score = torch.empty(B, N, 3, **f32)
        self.rigids = torch.empty(B, N, 7, **f32)
        self.atom37 = torch.empty(B, N, 37, 3, **f32)
        self.atom14 = torch.empty(B, N, 14, 3, **f32)
        self.trace_node = self.trace_edge = None
        if trace:
            self.trace_node = torch.zeros(d.num_blocks + 1, B, N, d.c_s, **f32)
            self.trace_edge = torch.zeros(d.num_blocks, B, N, N, d.c_z, **f32)
        self.trace_inner = (torch.zeros(d.num_blocks, 4, B, N, d.c_s + d.c_skip, **f32) if trace_inner else None)
        self.ev_start = self.ev_stop = None  # optional hipEvent pairs around the EdgeTransition launches (bench.py)
        self.score_table = None  # so3.use_cached_score: [B, num_omega] float64 rows of the score-norm table for this call's t
        self.omega_edges = None
        self.clock_out = None  # optional int64[3] device tensor: shader-clock probe of the EdgeTransition kernels (FdiptForwardArgs.clock_out)
        self.reserve_cus = 0  # CUs the persistent pair kernels leave to concurrent sub-batch streams (inference.StreamedLoops)
        self.t_emb_eps = torch.as_tensor(embedding.get_timestep_embedding(np.array([1e-5], dtype=np.float32), E)[0],
                                         device=dev)

    def forward(self, rigids_t, res_mask, fixed_mask, sc_ca_t, aatype, gt_psi, t_dev, t_emb_dev, sigma_dev,
                want_atoms: bool = True, ca_out=None, atom37_out=None, step_cursor=None):
        """All arguments are device tensors (float32 unless noted); outputs land in this state's buffers.
        ``ca_out`` ([B,N,3], may be ``sc_ca_t`` itself) receives the predicted CA positions for the next step.
        ``step_cursor`` (device int32[2], FdiptForwardArgs.step_cursor): ``rigids_t``, ``t_dev``, ``t_emb_dev``, ``sigma_dev``,
        ``self.score_table`` and ``atom37_out`` are then the step-major arrays of a trajectory and the kernels use row
        ``step_cursor[0]`` of each — the launch arguments no longer depend on the step (``inference.ReverseLoop``'s step graph)."""
        lib = _lib.load()
        net = self.net
        a = _lib.ForwardArgs()
        a.B, a.N, a.n_rel, a.rel_off = self.B, self.N, self.n_rel, self.rel_off
        for name, tns in (("rigids_t", rigids_t), ("res_mask", res_mask), ("fixed_mask", fixed_mask), ("sc_ca_t", sc_ca_t),
                          ("seq_idx", self.seq_idx), ("idx_emb", self.idx_emb), ("aatype", aatype), ("gt_psi", gt_psi),
                          ("t", t_dev), ("t_emb", t_emb_dev), ("t_emb_eps", self.t_emb_eps), ("so3_sigma", sigma_dev),
                          ("bb_tables", net.bb_tables), ("psi", self.psi), ("rot_score", self.rot_score),
                          ("trans_score", self.trans_score), ("rigids", self.rigids),
                          ("atom37", (self.atom37 if atom37_out is None else atom37_out) if want_atoms else None), ("atom14", self.atom14 if want_atoms else None),
                          ("trace_node", self.trace_node), ("trace_edge", self.trace_edge), ("trace_inner", self.trace_inner),
                          ("ca_out", ca_out)):
            setattr(a, name, _lib.ptr(tns))
        if self.ev_start is not None:
            a.ev_start, a.ev_stop = self.ev_start, self.ev_stop
        a.reserve_cus = self.reserve_cus
        a.step_cursor = _lib.ptr(step_cursor)
        a.clock_out = _lib.ptr(self.clock_out)
        if self.score_table is not None:
            a.so3_score_table, a.so3_omega_edges = _lib.ptr(self.score_table), _lib.ptr(self.omega_edges)
            a.so3_num_omega = int(self.score_table.shape[-1])
        _lib.check(lib.fdipt_score_forward(C.byref(net.dims), _lib.ptr(net.params), _lib.ptr(net.derived),
                                           _lib.ptr(self.setup), C.byref(a), _lib.ptr(self.ws), self.ws_bytes,
                                           _lib.stream_ptr()), "score_forward")


class ScoreNetwork:
    def __init__(self, model_conf, diffuser, inpainting: bool = False, precision: str = "fp32", device=None,
                 kernel_flags: int = 0):
        """``precision``: "fp32" (exact fp32 FMA chains, any width) or "fp16" (fp16 MFMA operands and pair representation,
        fp32 accumulation / frames / statistics: the throughput mode).  ``kernel_flags``: ``_lib.KF_*`` bits."""
        self._model_conf = model_conf
        self.diffuser = diffuser
        self.inpainting = inpainting
        self.precision = PRECISIONS[precision]
        self.dims = dims_from_conf(model_conf, diffuser._se3_conf, inpainting, self.precision, kernel_flags)
        self.shapes = weights.param_shapes(model_conf, inpainting)
        self.device = torch.device(device) if device is not None else None
        self.params = self.derived = self.bb_tables = None
        self._state = None
        self._state_key = None

    # ------------------------------------------------------------------ nn.Module-like surface used by inference.py
    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.FdiptError(f"ScoreNetwork runs only on the MI355X (requested {dev}); there is no CPU fallback")
        self.device = dev
        if self._host_params is not None:
            self._upload()
        return self

    def eval(self):
        return self

    _host_params = None

    def state_dict(self):
        flat = self._host_params
        out, o = OrderedDict(), 0
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            out[k] = flat[o:o + n].reshape(shp)
            o += n
        return out

    def load_state_dict(self, sd):
        """``sd``: name -> array/tensor with the reference's ``state_dict`` names (optional ``module.`` prefix,
        ``experiments/inference.py:156-159``)."""
        sd = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in sd.items()}
        missing = [k for k in self.shapes if k not in sd]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        parts = []
        for k, shp in self.shapes.items():
            v = sd[k]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            if tuple(v.shape) != tuple(shp):
                raise ValueError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shp)}")
            parts.append(v.astype(np.float32).reshape(-1))
        self._host_params = np.concatenate(parts)
        if self.device is not None:
            self._upload()
        return self

    def load_synthetic(self, seed: int = 0, bb_gain: float = weights.BB_GAIN):
        return self.load_state_dict(weights.synth_state_dict(self.shapes, seed, bb_gain))

    def _upload(self):
        lib = _lib.load()
        n = lib.fdipt_param_count(C.byref(self.dims))
        if n != len(self.shapes) or lib.fdipt_param_offset(C.byref(self.dims), n) != self._host_params.size:
            raise _lib.FdiptError("parameter inventory of libfdipt_hip disagrees with framedipt_amd.weights")
        with torch.cuda.device(self.device):
            self.params = torch.as_tensor(self._host_params, device=self.device)
            self.derived = torch.empty(lib.fdipt_derived_bytes(C.byref(self.dims)), dtype=torch.uint8, device=self.device)
            self.bb_tables = torch.as_tensor(residue_tables.packed_bytes(), device=self.device)
            _lib.check(lib.fdipt_model_prepare(C.byref(self.dims), _lib.ptr(self.params), _lib.ptr(self.derived),
                                               _lib.stream_ptr()), "model_prepare")
        self._state = None
        # step graphs captured by a live inference.ReverseLoop bake the old weight / derived-buffer pointers in: the loop compares this
        # counter with the one it recorded at capture time and drops its graphs when they differ
        self.weights_version = getattr(self, "weights_version", 0) + 1

    # ------------------------------------------------------------------ batch state
    def batch_state(self, seq_idx: torch.Tensor, trace: bool = False, trace_inner: bool = False) -> BatchState:
        if self.params is None:
            raise _lib.FdiptError("ScoreNetwork has no weights on a device: call load_state_dict(...) and .to('cuda')")
        key = (tuple(seq_idx.shape), seq_idx.detach().cpu().numpy().tobytes(), trace, trace_inner)
        if self._state is None or self._state_key != key:
            with torch.cuda.device(self.device):
                self._state = BatchState(self, seq_idx, trace, trace_inner)
            self._state_key = key
        return self._state

    def new_batch_state(self, seq_idx: torch.Tensor) -> BatchState:
        """A batch state (workspace + output buffers) of its own, not the cached one: sub-batches that run concurrently."""
        if self.params is None:
            raise _lib.FdiptError("ScoreNetwork has no weights on a device: call load_state_dict(...) and .to('cuda')")
        with torch.cuda.device(self.device):
            return BatchState(self, seq_idx)

    def step_scalars(self, t_host: np.ndarray):
        """Host-evaluated per-call inputs: float32 t, its timestep embedding, the IGSO(3) sigma (float64)."""
        t32 = np.asarray(t_host, dtype=np.float32).reshape(-1)
        temb = embedding.get_timestep_embedding(t32, self.dims.index_embed)
        sig = self.diffuser._so3_diffuser.score_sigma(t32)
        return t32, temb, sig

    # ------------------------------------------------------------------ forward (API-compatible)
    def __call__(self, input_feats: dict, trace: bool = False, trace_inner: bool = False) -> dict:
        dev = self.device
        rig = input_feats["rigids_t"]
        _lib.require_cuda(rig, "ScoreNetwork.forward")
        st = self.batch_state(input_feats["seq_idx"], trace, trace_inner)
        f32 = lambda x: x.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        fixed = f32(input_feats["fixed_mask"])
        aatype = preprocess_aatype(input_feats.get("aatype"), fixed, self.inpainting, self._model_conf.input_aatype)
        aatype_dev = None if aatype is None else aatype.to(device=dev, dtype=torch.int32).contiguous()
        t32, temb, sig = self.step_scalars(input_feats["t"].detach().cpu().numpy())
        gt = input_feats["torsion_angles_sin_cos"]
        so3 = self.diffuser._so3_diffuser
        if so3.use_cached_score:
            st.score_table = torch.as_tensor(so3.score_table_rows(t32), device=dev)
            st.omega_edges = torch.as_tensor(so3.omega_edges, device=dev)
        else:
            st.score_table = st.omega_edges = None
        with torch.cuda.device(dev):
            st.forward(f32(rig), f32(input_feats["res_mask"]), fixed, f32(input_feats["sc_ca_t"]), aatype_dev,
                       f32(gt[..., 2, :]), torch.as_tensor(t32, device=dev), torch.as_tensor(temb, device=dev),
                       torch.as_tensor(sig, device=dev))
        psi = st.psi.to(gt.dtype) if gt.dtype == torch.float64 else st.psi  # sn:259-260: psi inherits the gt dtype
        out = {"psi": psi.clone(), "rot_score": st.rot_score.clone(), "trans_score": st.trans_score.clone(),
               "rigids": st.rigids.clone(), "atom37": st.atom37.clone(), "atom14": st.atom14.clone()}
        if trace:
            out["trace_node"], out["trace_edge"] = st.trace_node, st.trace_edge
        if trace_inner:
            out["trace_inner"] = st.trace_inner
        return out
