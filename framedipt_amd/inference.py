"""Reverse-diffusion driver with the signature of ``experiments/utils.py:inference_fn`` (:511-626).

Python drives the loop; a step is two C-ABI calls (score-network forward, fused SE(3) reverse step) on one HIP stream, addressed
through a device-side step cursor so that a step is captured once as a HIP graph and replayed (``ReverseLoop``).  Nothing
crosses to the host inside the loop: the state (x_t, self-conditioning CA, trajectories) stays in HBM and the noise tape /
per-step scalars are uploaded up front (the reference does >= 5 device->host syncs per step, SURVEY.md section 0 finding 3).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, gpu_guard
from .model.score_network import preprocess_aatype


def draw_noise_tape(diffuser, n_steps: int, B: int, N: int):
    """Noise of the reverse steps from the global legacy ``np.random`` stream in the reference's order
    (per step: SO(3) draw ``so3_diffuser.py:590`` then R^3 draw ``r3_diffuser.py:371``, each [B,N,3])."""
    z_rot = np.zeros((n_steps, B, N, 3))
    z_trans = np.zeros((n_steps, B, N, 3))
    for s in range(n_steps):
        if diffuser._diffuse_rot:
            z_rot[s] = np.random.normal(size=(B, N, 3))
        if diffuser._diffuse_trans:
            z_trans[s] = np.random.normal(size=(B, N, 3))
    return z_rot, z_trans


def _backbone(net, n, t7, rot, trans, psi, aatype, atom37):
    lib = _lib.load()
    _lib.check(lib.fdipt_backbone_atoms(n, _lib.ptr(t7), _lib.ptr(rot), _lib.ptr(trans), _lib.ptr(psi), _lib.ptr(aatype),
                                        _lib.ptr(net.bb_tables), _lib.ptr(atom37), None, _lib.stream_ptr()),
               "backbone_atoms")


def get_atom_positions_from_rigids(model, rigids, psi_torsions, aatype=None) -> np.ndarray:
    """``experiments/utils.py:415-437``: backbone atom37 positions [B,N,37,3] (NumPy) of frames given as tensor_7 [B,N,7] (or a
    ``Rigid``), psi torsions [B,N,2] and residue types [B,N] (None: alanine), through the backbone kernel."""
    t7 = rigids.to_tensor_7() if hasattr(rigids, "to_tensor_7") else rigids
    dev = model.device
    with torch.cuda.device(dev):
        t7 = t7.to(device=dev, dtype=torch.float32).contiguous()
        psi = psi_torsions.to(device=dev, dtype=torch.float32).contiguous()
        aa = None if aatype is None else aatype.to(device=dev, dtype=torch.int32).contiguous()
        atom37 = torch.empty(t7.shape[:-1] + (37, 3), dtype=torch.float32, device=dev)
        _backbone(model, int(np.prod(t7.shape[:-1])), t7, None, None, psi, aa, atom37)
        return atom37.cpu().numpy()


class ReverseLoop:
    """Device-resident state of one batch of trajectories; ``run()``, or ``prime()`` then ``step(k)`` for k = 0..num_t-1.

    **Step graph (the default, ``graph=True``).**  Every reverse step but the last is the same ~70 kernel launches on different
    rows of the trajectory arrays.  The library addresses those rows through a device-side step cursor
    (``FdiptForwardArgs.step_cursor``, ``fdipt_se3_reverse_step_indexed``: the fused reverse step advances it), so the launch
    arguments of a step do not depend on the step: the first noisy step is enqueued eagerly through the cursor (it also runs every
    kernel once), the second one is captured as a HIP graph — and ``GRAPH_CHUNK`` consecutive steps as a second graph — and all
    later steps are replays: one ``hipGraphLaunch`` per chunk instead of ~67 ``hipLaunchKernel`` + 3 ctypes calls per step
    (0.09 ms against 0.24 ms of host work per 2.1 - 2.2 ms GPU step at N = 300, B = 8 on a quiet host; under any per-call overhead —
    a tracer costs 45 us per launch — only the replayed loop stays GPU-bound: profiles/r05_driver_cmd_repro.md).  Same kernels, same
    bits as the eager loop (``graph=False``; tests/test_gpu_round5.py, tools/soak_step_graph.py).  The graphs hold pointers into this
    loop's buffers and its ``BatchState``: they live and die with the loop."""

    GRAPH_CHUNK = 8  # steps per replay of the chunk graph (``run()``); single steps replay the one-step graph

    def __init__(self, model, diffuser, data_init, num_t, min_t, center=True, aux_traj=False, self_condition=True,
                 noise_scale=1.0, embed_self_conditioning=True, inpainting=False, input_aatype=False, noise_tape=None, state=None,
                 graph=True, verify=0):
        self.model, self.diffuser = model, diffuser
        dev = self.dev = model.device
        gpu_guard.check(dev, what="inference_fn")  # a foreign compute process on this GPU: warn / refuse (FDIPT_SHARED_GPU)
        self.verify, self.verified = int(verify or 0), 0
        rig0 = data_init["rigids_t"]
        _lib.require_cuda(rig0, "inference_fn")
        if rig0.dim() == 2:
            raise ValueError("rigids_t needs a leading batch dimension")
        B, N = self.B, self.N = rig0.shape[0], rig0.shape[1]
        f32 = lambda x: x.to(device=dev, dtype=torch.float32).contiguous().clone()  # noqa: E731
        self.res_mask, self.fixed = f32(data_init["res_mask"]), f32(data_init["fixed_mask"])
        self.fixed_mask = (self.fixed * self.res_mask).contiguous()
        self.diffuse_mask = ((1 - self.fixed) * self.res_mask).contiguous()
        # two aatype views, as in the reference: the network pre-processes with ITS OWN flags (score_network.py:226-232:
        # self.inpainting / model_conf.input_aatype), the atom37 frames of the trajectory with inference_fn's arguments
        # (experiments/utils.py:376-388)
        to_dev = lambda x: None if x is None else x.to(device=dev, dtype=torch.int32).contiguous()  # noqa: E731
        self._inpainting, self._input_aatype = inpainting, input_aatype
        self.aatype = to_dev(preprocess_aatype(data_init.get("aatype"), self.fixed, inpainting, input_aatype))
        self.net_aatype = to_dev(preprocess_aatype(data_init.get("aatype"), self.fixed, model.inpainting,
                                                   model._model_conf.input_aatype))
        # rigid_0_traj (bb_0_pred, experiments/utils.py:397-402) is built with inference_fn's aatype.  Where that equals the
        # network's own view the forward writes the row itself; where it does not (the reference's default inpainting
        # configuration: inference.input_aatype=True with model.input_aatype=False, so the network sees 20 = unknown on the
        # diffused residues and would build ALA there) the row comes from a backbone launch on the forward's frames / psi
        same = (self.aatype is None and self.net_aatype is None) or (
            self.aatype is not None and self.net_aatype is not None and bool(torch.equal(self.aatype, self.net_aatype)))
        # (net_aatype None = the backbone builder's default residue 0, as is aatype None)
        self.bb0_from_forward = same
        self.gt_tors = data_init["torsion_angles_sin_cos"]
        self.gt_psi = f32(self.gt_tors[..., 2, :])
        self.st = state if state is not None else model.batch_state(data_init["seq_idx"])
        self.num_t, self.min_t, self.dt = num_t, min_t, 1 / num_t
        self.center, self.aux_traj, self.noise_scale = center, aux_traj, noise_scale
        self.self_condition, self.embed_sc = self_condition, embed_self_conditioning
        self.reverse_steps = np.linspace(min_t, 1.0, num_t)[::-1]
        # steps that take the reverse SDE step (t > min_t): a prefix of the schedule (it decreases to min_t)
        n_noisy = self.n_noisy = int(np.sum(self.reverse_steps > min_t))
        t32, temb, sig = model.step_scalars(self.reverse_steps)
        with torch.cuda.device(dev):
            self.t_all = torch.as_tensor(np.repeat(t32[:, None], B, 1), device=dev)
            self.temb_all = torch.as_tensor(np.repeat(temb[:, None, :], B, 1), device=dev)
            self.sig_all = torch.as_tensor(np.repeat(sig[:, None], B, 1), device=dev)
            self.t_tab = torch.as_tensor(np.ascontiguousarray(self.reverse_steps, dtype=np.float64), device=dev)
            so3 = diffuser._so3_diffuser
            self.tab_all = self.omega_edges = None
            if so3.use_cached_score:  # one row of the score-norm table per step (all samples of a batch share t)
                rows = so3.score_table_rows(t32)
                self.tab_all = torch.as_tensor(np.repeat(rows[:, None, :], B, 1), device=dev)
                self.omega_edges = torch.as_tensor(so3.omega_edges, device=dev)
            if noise_tape is None:
                noise_tape = draw_noise_tape(diffuser, n_noisy, B, N)
            self.z_rot = torch.as_tensor(np.ascontiguousarray(noise_tape[0], dtype=np.float64), device=dev)
            self.z_trans = torch.as_tensor(np.ascontiguousarray(noise_tape[1], dtype=np.float64), device=dev)
            self.sc_ca = f32(data_init["sc_ca_t"])
            self.rigid_traj = torch.empty(num_t + 1, B, N, 7, device=dev)  # row k = x_t of step k (row 0 = x_T), row k + 1 = its x_{t-1}
            self.rigid_traj[0] = rig0.to(device=dev, dtype=torch.float32)
            self.prot_traj = torch.empty(num_t, B, N, 37, 3, device=dev)
            self.bb0_traj = torch.empty(num_t, B, N, 37, 3, device=dev) if aux_traj else None
            self.trans_traj = torch.empty(num_t, B, N, 3, device=dev) if aux_traj else None
            self.cursor = torch.zeros(2, dtype=torch.int32, device=dev)  # FdiptForwardArgs.step_cursor: {step index, ticket}
        # step graph: needs the noise rows of step k at row k (any tape drawn for this schedule), and at least a few steps to pay for capture
        self.graph = bool(graph) and n_noisy >= 3 and self.z_rot.shape[0] >= n_noisy and self.z_trans.shape[0] >= n_noisy
        self._cursor_host = 0      # host mirror of cursor[0] (every writer of the cursor is ordered on this loop's stream)
        self._g1 = self._gn = None  # captured one-step / GRAPH_CHUNK-step graphs
        self.capture_seconds = 0.0

    @property
    def rigids_t(self):
        """x_t of the next step to run (the row of the rigid trajectory the last finished step wrote)."""
        return self.rigid_traj[self._next]

    _next = 0

    # ------------------------------------------------------------------ eager launches (pointers of step k on the host side)
    def _fwd(self, k, want_atoms, sc_update):
        # the forward itself hands the predicted CA positions to the next step's self-conditioning input (read at its start,
        # written at its end: no copy kernel)
        direct = want_atoms and self.bb0_from_forward
        self.st.score_table = None if self.tab_all is None else self.tab_all[k]
        self.st.omega_edges = self.omega_edges
        self.st.forward(self.rigid_traj[k], self.res_mask, self.fixed, self.sc_ca, self.net_aatype, self.gt_psi, self.t_all[k],
                        self.temb_all[k], self.sig_all[k], direct, ca_out=self.sc_ca if sc_update else None,
                        atom37_out=self.bb0_traj[k] if direct else None)  # rigid_0_traj row: straight into its slot
        if want_atoms and not direct:
            _backbone(self.model, self.B * self.N, self.st.rigids, None, None, self.st.psi, self.aatype, self.bb0_traj[k])

    def prime(self):
        """Self-conditioning priming call (utils.py:571-578)."""
        if self.embed_sc and self.self_condition:
            with torch.cuda.device(self.dev):
                self._fwd(0, False, True)

    def _verify_forward(self, k):
        """verify=n: the forward of step k runs twice on the same inputs (the first time without its side effects: no self-conditioning
        hand-over, no trajectory row) and every output must come back bit-identical — cheap insurance against silent corruption by a
        foreign process on the GPU.  One host synchronisation per verified step."""
        st = self.st
        self.st.score_table = None if self.tab_all is None else self.tab_all[k]
        self.st.omega_edges = self.omega_edges
        st.forward(self.rigid_traj[k], self.res_mask, self.fixed, self.sc_ca, self.net_aatype, self.gt_psi, self.t_all[k],
                   self.temb_all[k], self.sig_all[k], False)
        return [x.clone() for x in (st.rigids, st.psi, st.rot_score, st.trans_score)]

    def _step_eager(self, k):
        st, t, n = self.st, self.reverse_steps[k], self.B * self.N
        first = self._verify_forward(k) if self.verify and k % self.verify == 0 else None
        self._fwd(k, self.aux_traj, self.embed_sc and t > self.min_t)
        if first is not None:
            for name, a, b in zip(("rigids", "psi", "rot_score", "trans_score"), first, (st.rigids, st.psi, st.rot_score, st.trans_score)):
                if not torch.equal(a, b):
                    raise _lib.FdiptError(f"verify: two runs of the forward of step {k} differ in `{name}` (max |diff| "
                                          f"{float((a - b).abs().max()):.3e}): results on this GPU are not reproducible — is another "
                                          "compute process using it? (DESIGN.md section 6)")
            self.verified += 1
        nxt = self.rigid_traj[k + 1]
        if t > self.min_t:
            # x_{t-1} lands in its trajectory slot, which is the next forward's input; its atom37 frame comes out of the
            # same launch
            self.diffuser.reverse_device(self.rigid_traj[k], st.rot_score, st.trans_score, self.diffuse_mask,
                                         self.z_rot[k], self.z_trans[k], t, self.dt, self.center,
                                         self.noise_scale, rigids_out=nxt,
                                         atoms=(st.psi, self.aatype, self.model.bb_tables, self.prot_traj[k]),
                                         traj=(st.rigids, self.fixed_mask, self.trans_traj[k]) if self.aux_traj else None)
        else:  # last step: take the x_0 prediction, utils.py:373-374
            nxt.copy_(st.rigids)
            _backbone(self.model, n, nxt, None, None, st.psi, self.aatype, self.prot_traj[k])
            if self.aux_traj:  # (on the other steps the reverse-step launch writes this row)
                self.trans_traj[k] = self.diffuse_mask[..., None] * st.rigids[..., 4:] + self.fixed_mask[..., None] * nxt[..., 4:]

    # ------------------------------------------------------------------ cursor-addressed launches (the same for every noisy step)
    def _enqueue_indexed(self):
        """One noisy step (forward, rigid_0_traj row, fused reverse step) on the rows ``cursor[0]`` of the trajectory arrays; the
        reverse-step launch advances the cursor.  Nothing here depends on the step: this is what the step graphs hold."""
        lib, st, d = _lib.load(), self.st, self.diffuser
        direct = self.aux_traj and self.bb0_from_forward
        st.score_table, st.omega_edges = self.tab_all, self.omega_edges
        st.forward(self.rigid_traj, self.res_mask, self.fixed, self.sc_ca, self.net_aatype, self.gt_psi, self.t_all, self.temb_all,
                   self.sig_all, direct, ca_out=self.sc_ca if self.embed_sc else None, atom37_out=self.bb0_traj if direct else None,
                   step_cursor=self.cursor)
        if self.aux_traj and not direct:
            _lib.check(lib.fdipt_backbone_atoms_indexed(self.B * self.N, _lib.ptr(st.rigids), _lib.ptr(st.psi), _lib.ptr(self.aatype),
                                                        _lib.ptr(self.model.bb_tables), _lib.ptr(self.bb0_traj), _lib.ptr(self.cursor),
                                                        _lib.stream_ptr()), "backbone_atoms_indexed")
        so3, r3 = d._so3_diffuser, d._r3_diffuser
        a = _lib.ReverseIndexed()
        a.B, a.N = self.B, self.N
        for name, tns in (("rigid_traj", self.rigid_traj), ("rot_score", st.rot_score), ("trans_score", st.trans_score),
                          ("diffuse_mask", self.diffuse_mask), ("z_rot", self.z_rot), ("z_trans", self.z_trans), ("t_table", self.t_tab),
                          ("psi", st.psi), ("aatype", self.aatype), ("bb_tables", self.model.bb_tables), ("prot_traj", self.prot_traj),
                          ("pred_rigids", st.rigids if self.aux_traj else None), ("traj_fixed_mask", self.fixed_mask if self.aux_traj else None),
                          ("trans_traj", self.trans_traj), ("step_cursor", self.cursor)):
            setattr(a, name, _lib.ptr(tns))
        a.dt, a.noise_scale = float(self.dt), float(self.noise_scale)
        a.center, a.diffuse_rot, a.diffuse_trans = int(bool(self.center)), int(bool(d._diffuse_rot)), int(bool(d._diffuse_trans))
        a.so3_min_sigma, a.so3_max_sigma, a.r3_min_b, a.r3_max_b = so3.min_sigma, so3.max_sigma, r3.min_b, r3.max_b
        a.coordinate_scaling = r3._r3_conf.coordinate_scaling
        _lib.check(lib.fdipt_se3_reverse_step_indexed(C.byref(a), _lib.stream_ptr()), "se3_reverse_step_indexed")

    def capture_step_graph(self, n_steps=1):
        """A HIP graph of ``n_steps`` consecutive cursor-addressed steps (nothing runs; whatever ``self.st`` carries — event pairs,
        the clock probe — is captured with it).  Capture happens on a side stream: the caller's stream may be the null stream."""
        import time
        t0 = time.perf_counter()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.dev)
        with torch.cuda.stream(side):
            g.capture_begin(capture_error_mode="thread_local")
            try:
                for _ in range(n_steps):
                    self._enqueue_indexed()
            finally:
                g.capture_end()
        self.capture_seconds += time.perf_counter() - t0
        self._captured_weights = getattr(self.model, "weights_version", 0)  # the graphs bake the weight / derived-buffer pointers in
        return g

    def _set_cursor(self, k):
        if self._cursor_host != k:
            self.cursor[:1].fill_(k)
            self._cursor_host = k

    # (device, precision, kernel flags, B, N) of the cursor-addressed steps that have run in this process: the kernels' lazy per-device
    # set-up (raised dynamic-LDS caps) must not fall into a capture, so the first such step of a shape class is enqueued eagerly
    _WARM: set = set()

    def _warm_key(self):
        # every field of FdiptDims (widths select the fused or the generic kernels) and the options that change which kernels a step
        # launches (round-5 advisor: a second model with other widths at the same B, N was treated as warm)
        return (str(self.dev), bytes(self.model.dims), self.B, self.N, bool(self.aux_traj), bool(self.bb0_from_forward),
                bool(self._inpainting), self.tab_all is not None, bool(self.embed_sc), bool(self.self_condition))

    def prepare(self):
        """Capture the step graphs now (otherwise: lazily at the first replay).  Nothing runs on the GPU."""
        if self.graph and self._warm_key() in self._WARM:
            with torch.cuda.device(self.dev):
                if self._g1 is None:
                    self._g1 = self.capture_step_graph(1)
                if self._gn is None and self.GRAPH_CHUNK > 1 and self.n_noisy >= 2 * self.GRAPH_CHUNK:
                    self._gn = self.capture_step_graph(self.GRAPH_CHUNK)
        return self

    def _advance(self, k, n, eager=False):
        """Steps k .. k+n-1 (all noisy) through the cursor: launch by launch the first time a shape class runs in the process (or with
        ``eager``: bench.py's event-bracketed steps — events recorded inside a graph cannot be timed on ROCm), graph replays otherwise."""
        self._set_cursor(k)
        if (self._g1 is not None or self._gn is not None) and self._captured_weights != getattr(self.model, "weights_version", 0):
            self._g1 = self._gn = None  # the model's weights were (re)loaded behind this loop: its captured graphs point at freed buffers
        if eager or self._warm_key() not in self._WARM:
            for _ in range(n):
                self._enqueue_indexed()
            self._WARM.add(self._warm_key())
        elif n > 1 and n == self.GRAPH_CHUNK:
            if self._gn is None:
                self._gn = self.capture_step_graph(n)
            self._gn.replay()
        else:
            if self._g1 is None:
                self._g1 = self.capture_step_graph(1)
            for _ in range(n):
                self._g1.replay()
        self._cursor_host = k + n

    def step(self, k, eager=False):
        """one_step_inference (utils.py:292-412) for reverse step k."""
        with torch.cuda.device(self.dev):
            if self.graph and k < self.n_noisy and not (self.verify and k % self.verify == 0):
                self._advance(k, 1, eager)
            else:
                self._step_eager(k)
        self._next = k + 1

    def run_steps(self, k_begin, k_end, eager_steps=(), before_step=None):
        """Steps k_begin .. k_end - 1 in sequence (graph replays, ``GRAPH_CHUNK`` steps at a time where that many noisy steps remain).
        ``eager_steps``: steps to enqueue launch by launch instead, ``before_step(k)`` called ahead of each of them and ``before_step(None)``
        behind it (bench.py: HIP events around the dominant kernel's launches of a few steps)."""
        k, eager_steps = k_begin, set(eager_steps)
        while k < k_end:
            n = self.GRAPH_CHUNK
            if (not self.graph or n < 2 or min(self.n_noisy, k_end) - k < n or self._warm_key() not in self._WARM
                    or any(kk in eager_steps or (self.verify and kk % self.verify == 0) for kk in range(k, k + n))):
                n = 1
            if n == 1:
                if k in eager_steps and before_step is not None:
                    before_step(k)
                self.step(k, eager=k in eager_steps)
                if k in eager_steps and before_step is not None:
                    before_step(None)
            else:
                with torch.cuda.device(self.dev):
                    self._advance(k, n)
            k += n
        self._next = k_end
        return self

    def run(self, eager_steps=(), before_step=None):
        """The whole trajectory: priming forward, then every step."""
        self.prime()
        return self.run_steps(0, self.num_t, eager_steps, before_step)

    def results(self, return_device=False):
        st = self.st
        psi_pred = st.psi.to(self.gt_tors.dtype).clone() if self.gt_tors.dtype == torch.float64 else st.psi.clone()
        conv = (lambda x: torch.flip(x, (0,))) if return_device else (lambda x: np.flip(x.cpu().numpy(), (0,)))
        ret = {"prot_traj": conv(self.prot_traj)}
        if self.aux_traj:
            ret["rigid_traj"] = conv(self.rigid_traj)
            ret["trans_traj"] = conv(self.trans_traj)
            ret["psi_pred"] = psi_pred[None]
            ret["rigid_0_traj"] = conv(self.bb0_traj)
        return ret


_SUB_BATCH_STREAMS: dict = {}  # device -> HIP streams of the sub-batches, created once (HIP multiplexes streams onto a few hardware
                               # queues: fresh streams per trajectory would end up sharing one)


def _sub_batch_streams(dev, n):
    pool = _SUB_BATCH_STREAMS.setdefault(str(dev), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


class StreamedLoops:
    """The batch cut into ``n_streams`` sub-batches, each a ``ReverseLoop`` with its own workspace on its own HIP stream.

    The node path of the score network (M = B*N rows: a few dozen workgroups per launch) is latency-bound and leaves most of the
    256 CUs idle, while the pair path (EdgeTransition, IPA attention, o_pair) is throughput-bound: with two sub-batches in
    flight the GPU runs one sub-batch's pair kernels under the other's node-path launches.  Results do not change: a sample's
    trajectory is bit-identical whatever batch it rides in (tests: test_batch_of_equal_samples_matches_single,
    test_concurrent_forwards_are_bit_identical).  Same interface as ``ReverseLoop`` (prime / step / results)."""

    MAX_STREAMS = 2   # experimental feature, verified for two streams only (see below)
    MAX_LENGTH = 384  # ... and for N <= 384 only
    OPT_IN_ENV = "FDIPT_EXPERIMENTAL_STREAMS"  # more than one stream is an explicit opt-in: experimental=True or this variable set to 1

    def __init__(self, model, diffuser, data_init, n_streams, num_t, min_t, noise_tape=None, reserve_cus=48, experimental=False, **kw):
        # Soak results (tools/soak_streams.sh, tools/streams_stat.py; trajectories against the single-stream run):
        #   * N = 128 / 300, two streams, any number of reserved CUs: 0 mismatching runs of ~400;
        #   * N = 300, three or four streams: 15 - 40 % of the runs differ in one sample from some step on;
        #   * N = 512 / 724 / 1000, two streams: 1 / 12, 3 / 12, up to 11 / 12 mismatching runs depending on `reserve_cus`.
        # Cause (round 4): on these GPUs a wave that runs a half-precision MFMA and lane-masked VALU code disturbs OTHER
        # waves on its SIMD (their last 16-lane pass is written with the foreign EXEC mask) — reproduced without library code by
        # tools/micro/hazard_repro.hip.  It needs waves of different kernels on one SIMD, which only concurrent streams / processes
        # produce; inside the verified range the kernels' footprints happen to keep the small fp64 kernels off the attention's SIMDs.
        # Nothing in software removes it, so: an explicit opt-in, refused beyond the range the soaks cover.  The single-stream path
        # is not exposed and is bit-reproducible at every size.
        import os
        B = data_init["rigids_t"].shape[0]
        requested, n_streams = n_streams, max(1, min(n_streams, B))  # (the limits apply to what would actually run)
        if n_streams > self.MAX_STREAMS:
            raise ValueError(f"streams={requested}: sub-batch streams are verified bit-identical to the single-stream run for at most "
                             f"{self.MAX_STREAMS} streams")
        if n_streams > 1 and data_init["rigids_t"].shape[1] > self.MAX_LENGTH:
            raise ValueError(f"streams={requested} at N = {data_init['rigids_t'].shape[1]}: sub-batch streams are verified bit-identical to "
                             f"the single-stream run for N <= {self.MAX_LENGTH} only")
        if n_streams > 1 and not (experimental or os.environ.get(self.OPT_IN_ENV) == "1"):
            raise ValueError(f"streams={requested}: concurrent sub-batch streams are an experimental feature with an open correctness item beyond "
                             f"the verified range (DESIGN.md, concurrency): opt in with experimental_streams=True or {self.OPT_IN_ENV}=1")
        cuts = [round(i * B / n_streams) for i in range(n_streams + 1)]
        self.dev = model.device
        if noise_tape is None:
            n_noisy = int(np.sum(np.linspace(min_t, 1.0, num_t)[::-1] > min_t))
            noise_tape = draw_noise_tape(diffuser, n_noisy, B, data_init["rigids_t"].shape[1])
        self.loops, self.streams = [], _sub_batch_streams(self.dev, n_streams)
        with torch.cuda.device(self.dev):
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                sub = {k: v[lo:hi] for k, v in data_init.items() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B}
                st = model.new_batch_state(sub["seq_idx"])
                st.reserve_cus = reserve_cus if n_streams > 1 else 0  # the persistent pair kernels leave CUs to the other streams
                self.loops.append(ReverseLoop(model, diffuser, sub, num_t, min_t, noise_tape=tuple(z[:, lo:hi] for z in noise_tape),
                                              state=st, graph=False, **kw))
        torch.cuda.synchronize(self.dev)  # set-up ran on the current stream
        self.st = self.loops[0].st

    def __del__(self):
        # the loops' buffers were allocated on the caller's stream and are used on the side streams: if the object is dropped
        # early (an exception mid-trajectory) the caching allocator must not hand them out while side-stream kernels still run
        try:
            self.synchronize()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _each(self, fn):
        # (no per-step ordering against the caller's stream: an event on the default stream orders every sub-batch stream behind
        #  all work enqueued before it, i.e. the sub-batches would run one after the other; set-up was synchronised once)
        for loop, s in zip(self.loops, self.streams):
            with torch.cuda.stream(s):
                fn(loop)

    def prime(self):
        self._each(lambda lp: lp.prime())

    def step(self, k):
        self._each(lambda lp: lp.step(k))

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def results(self, return_device=False):
        self.synchronize()
        parts = [lp.results(return_device) for lp in self.loops]
        cat = lambda xs: torch.cat(xs, 1) if torch.is_tensor(xs[0]) else np.concatenate(xs, 1)  # noqa: E731
        return {k: cat([p[k] for p in parts]) for k in parts[0]}  # (every returned array carries the batch on axis 1)


def inference_fn(model, diffuser, data_init, num_t, min_t, center=True, aux_traj=False, self_condition=True,
                 noise_scale=1.0, embed_self_conditioning=True, inpainting=False, input_aatype=False, noise_tape=None,
                 return_device=False, streams=1, experimental_streams=False, graph=True, verify=0, pad_to_four=True):
    """Same arguments / returned keys as the reference.  ``noise_tape=(z_rot, z_trans)`` ([num_t-1,B,N,3] float64
    N(0,1) draws) overrides the global ``np.random`` stream (sample-sharded runs).  ``data_init`` tensors carry a
    leading batch dimension B >= 1 (the reference always passes B = 1).  ``graph=True`` (default): the steps are replays of a HIP
    graph captured once per trajectory (``ReverseLoop``: same kernels and bits, ~1/70 of the host work); ``graph=False``: every step
    enqueued launch by launch.  ``verify=k``: the forward of every k-th step runs twice and must reproduce its bits (``FdiptError``
    otherwise; one host sync per verified step) — for GPUs shared with other compute processes (``gpu_guard``).  ``streams=n``: the batch runs as n sub-batches on n HIP streams (same results; the latency-bound
    node path of one sub-batch overlaps the pair kernels of the other) — experimental: needs ``experimental_streams=True`` (or
    FDIPT_EXPERIMENTAL_STREAMS=1), at most two streams, N <= 384 (``StreamedLoops``; eager launches)."""
    # Round 6: lengths that are no multiple of 4 run the half-precision mode's fall-back pair kernels (edge_transition3, the pass over z for
    # o_pair): 2.94 ms per step at N = 302 against 2.14 at 304 (eight samples).  ``pad_to_four`` (default) pads such a sample with masked rows
    # (res_mask = 0, identity frames, zero noise: sharding.pad_item, what run_sharded does to mixed-length batches) and cuts the returned
    # arrays back to N — the real residues see the same arithmetic as in a run_sharded batch of their kernel class.  The noise tape is drawn
    # for the REAL residues first (the reference's np.random order is untouched).
    n_real = int(data_init["rigids_t"].shape[1])
    n_pad = -(-n_real // 4) * 4
    padded = bool(pad_to_four) and n_pad != n_real and getattr(model, "precision", _lib.PREC_F32) != _lib.PREC_F32
    if padded:
        from . import sharding
        if noise_tape is None:
            n_noisy = int(np.sum(np.linspace(min_t, 1.0, num_t)[::-1] > min_t))
            noise_tape = draw_noise_tape(diffuser, n_noisy, int(data_init["rigids_t"].shape[0]), n_real)
        data_init, noise_tape = sharding.pad_item(data_init, noise_tape, n_pad)
    if streams > 1 and data_init["rigids_t"].shape[0] > 1:  # sub-batches on their own HIP streams (same results)
        # (the sub-loops are eager ReverseLoops: verify= is passed through to them, graph= does not apply)
        loop = StreamedLoops(model, diffuser, data_init, streams, num_t, min_t, noise_tape=noise_tape, center=center, aux_traj=aux_traj,
                             self_condition=self_condition, noise_scale=noise_scale, embed_self_conditioning=embed_self_conditioning,
                             inpainting=inpainting, input_aatype=input_aatype, experimental=experimental_streams, verify=verify)
        loop.prime()
        for k in range(num_t):
            loop.step(k)
    else:
        loop = ReverseLoop(model, diffuser, data_init, num_t, min_t, center, aux_traj, self_condition, noise_scale,
                           embed_self_conditioning, inpainting, input_aatype, noise_tape, graph=graph, verify=verify).run()
    res = loop.results(return_device)
    if padded:  # (every returned array carries the residues on the axis behind the batch)
        res = {k: v[:, :, :n_real] for k, v in res.items()}
    return res
