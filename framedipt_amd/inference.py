"""Reverse-diffusion driver with the signature of ``experiments/utils.py:inference_fn`` (:511-626).

Python drives the loop; every step is two C-ABI calls (score-network forward, fused SE(3) reverse step) plus the
backbone-atom kernel, all enqueued on one HIP stream.  Nothing crosses to the host inside the loop: the state
(x_t, self-conditioning CA, trajectories) stays in HBM and the noise tape / per-step scalars are uploaded up front
(the reference does >= 5 device->host syncs per step, SURVEY.md section 0 finding 3).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
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
    """Device-resident state of one batch of trajectories; ``prime()`` then ``step(k)`` for k = 0..num_t-1."""

    def __init__(self, model, diffuser, data_init, num_t, min_t, center=True, aux_traj=False, self_condition=True,
                 noise_scale=1.0, embed_self_conditioning=True, inpainting=False, input_aatype=False, noise_tape=None, state=None):
        self.model, self.diffuser = model, diffuser
        dev = self.dev = model.device
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
        n_noisy = int(np.sum(self.reverse_steps > min_t))
        t32, temb, sig = model.step_scalars(self.reverse_steps)
        with torch.cuda.device(dev):
            self.t_all = torch.as_tensor(np.repeat(t32[:, None], B, 1), device=dev)
            self.temb_all = torch.as_tensor(np.repeat(temb[:, None, :], B, 1), device=dev)
            self.sig_all = torch.as_tensor(np.repeat(sig[:, None], B, 1), device=dev)
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
            self.rigids_t = f32(rig0)
            self.sc_ca = f32(data_init["sc_ca_t"])
            self.rigid_traj = torch.empty(num_t + 1, B, N, 7, device=dev)
            self.rigid_traj[0] = self.rigids_t
            self.prot_traj = torch.empty(num_t, B, N, 37, 3, device=dev)
            self.bb0_traj = torch.empty(num_t, B, N, 37, 3, device=dev) if aux_traj else None
            self.trans_traj = torch.empty(num_t, B, N, 3, device=dev) if aux_traj else None
        self.noisy = 0

    def _fwd(self, k, want_atoms, sc_update):
        # the forward itself hands the predicted CA positions to the next step's self-conditioning input (read at its start,
        # written at its end: no copy kernel)
        direct = want_atoms and self.bb0_from_forward
        self.st.score_table = None if self.tab_all is None else self.tab_all[k]
        self.st.omega_edges = self.omega_edges
        self.st.forward(self.rigids_t, self.res_mask, self.fixed, self.sc_ca, self.net_aatype, self.gt_psi, self.t_all[k],
                        self.temb_all[k], self.sig_all[k], direct, ca_out=self.sc_ca if sc_update else None,
                        atom37_out=self.bb0_traj[k] if direct else None)  # rigid_0_traj row: straight into its slot
        if want_atoms and not direct:
            _backbone(self.model, self.B * self.N, self.st.rigids, None, None, self.st.psi, self.aatype, self.bb0_traj[k])

    def prime(self):
        """Self-conditioning priming call (utils.py:571-578)."""
        if self.embed_sc and self.self_condition:
            with torch.cuda.device(self.dev):
                self._fwd(0, False, True)

    def step(self, k):
        """one_step_inference (utils.py:292-412) for reverse step k."""
        st, t, n = self.st, self.reverse_steps[k], self.B * self.N
        with torch.cuda.device(self.dev):
            self._fwd(k, self.aux_traj, self.embed_sc and t > self.min_t)
            nxt = self.rigid_traj[k + 1]
            if t > self.min_t:
                # x_{t-1} lands in its trajectory slot, which is the next forward's input; its atom37 frame comes out of the
                # same launch
                self.diffuser.reverse_device(self.rigids_t, st.rot_score, st.trans_score, self.diffuse_mask,
                                             self.z_rot[self.noisy], self.z_trans[self.noisy], t, self.dt, self.center,
                                             self.noise_scale, rigids_out=nxt,
                                             atoms=(st.psi, self.aatype, self.model.bb_tables, self.prot_traj[k]),
                                             traj=(st.rigids, self.fixed_mask, self.trans_traj[k]) if self.aux_traj else None)
                self.noisy += 1
            else:  # last step: take the x_0 prediction, utils.py:373-374
                nxt.copy_(st.rigids)
                _backbone(self.model, n, nxt, None, None, st.psi, self.aatype, self.prot_traj[k])
                if self.aux_traj:  # (on the other steps the reverse-step launch writes this row)
                    self.trans_traj[k] = self.diffuse_mask[..., None] * st.rigids[..., 4:] + self.fixed_mask[..., None] * nxt[..., 4:]
            self.rigids_t = nxt

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
                                              state=st, **kw))
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


class GraphedTrajectory:
    """A whole trajectory — priming forward + ``num_t`` reverse steps, ~70 kernel launches each — captured ONCE as a HIP graph and
    replayed for every later batch of the same shape (``inference_fn(graph=True)``).

    Today the step is kernel-bound (the host needs 1.45 ms to enqueue a 2.25 ms step at N = 300, B = 8: ``tools/graph_step.py``), so a
    replay is not faster than the eager loop; the option exists so that the Python / ctypes enqueue cost (65 % of the GPU step) cannot
    become the bound as the kernels get faster, and for hosts with slow or busy cores.  Everything a step touches lives in buffers of
    the ``ReverseLoop`` this object owns (state, masks, noise tape, per-step scalars, trajectories), so a new batch is loaded by
    copying its inputs INTO those buffers (``load``), never by rebinding them.  Same results as the eager loop, bit for bit
    (tests/test_gpu_round4.py::test_graphed_trajectory_matches_the_eager_loop).  Capture needs every kernel of the trajectory to have
    run once in the process (lazy per-device launch attributes): the first ``graph=True`` call of a shape therefore runs eagerly and the
    second one captures."""

    def __init__(self, model, diffuser, data_init, num_t, min_t, noise_tape, **kw):
        self.dev = model.device
        with torch.cuda.device(self.dev):
            self.loop = lp = ReverseLoop(model, diffuser, data_init, num_t, min_t, noise_tape=noise_tape,
                                         state=model.new_batch_state(data_init["seq_idx"]), **kw)
            self.x_T = lp.rigids_t  # (the tensor the first step reads; ReverseLoop.step rebinds the attribute, not the buffer)
            self.graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # (allocator warm-up for the few torch ops of the last step)
                tmp = lp.diffuse_mask[..., None] * lp.st.rigids[..., 4:] + lp.fixed_mask[..., None] * lp.rigid_traj[1][..., 4:]
                del tmp
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(self.graph):
                lp.prime()
                for k in range(num_t):
                    lp.step(k)
        self.replays = 0

    def load(self, data_init, noise_tape):
        """Copy a new batch's inputs into the captured buffers (same B, N, seq_idx, aatype presence as at capture)."""
        lp, dev = self.loop, self.dev
        f32 = lambda x: x.to(device=dev, dtype=torch.float32)  # noqa: E731
        with torch.cuda.device(dev):
            res_mask, fixed = f32(data_init["res_mask"]), f32(data_init["fixed_mask"])
            lp.res_mask.copy_(res_mask)
            lp.fixed.copy_(fixed)
            lp.fixed_mask.copy_(fixed * res_mask)
            lp.diffuse_mask.copy_((1 - fixed) * res_mask)
            for mine, (inp, ia) in ((lp.aatype, (lp._inpainting, lp._input_aatype)),
                                    (lp.net_aatype, (lp.model.inpainting, lp.model._model_conf.input_aatype))):
                new = preprocess_aatype(data_init.get("aatype"), fixed, inp, ia)
                if (mine is None) != (new is None):
                    raise ValueError("graph replay: the batch differs from the captured one in whether residue types are given")
                if mine is not None:
                    mine.copy_(new.to(device=dev, dtype=torch.int32))
            same = (lp.aatype is None and lp.net_aatype is None) or (
                lp.aatype is not None and lp.net_aatype is not None and bool(torch.equal(lp.aatype, lp.net_aatype)))
            if same != lp.bb0_from_forward:  # (decides whether rigid_0_traj rows come from the forward or from a backbone launch)
                raise ValueError("graph replay: the batch differs from the captured one in how its x_0 backbone rows are built")
            lp.gt_tors = data_init["torsion_angles_sin_cos"]
            lp.gt_psi.copy_(f32(lp.gt_tors[..., 2, :]))
            self.x_T.copy_(f32(data_init["rigids_t"]))
            lp.rigid_traj[0].copy_(self.x_T)
            lp.sc_ca.copy_(f32(data_init["sc_ca_t"]))
            lp.z_rot.copy_(torch.as_tensor(np.ascontiguousarray(noise_tape[0], dtype=np.float64), device=dev))
            lp.z_trans.copy_(torch.as_tensor(np.ascontiguousarray(noise_tape[1], dtype=np.float64), device=dev))

    def run(self, data_init=None, noise_tape=None, return_device=False):
        if data_init is not None:
            self.load(data_init, noise_tape)
        with torch.cuda.device(self.dev):
            self.graph.replay()
        self.replays += 1
        return self.loop.results(return_device)


def _graph_key(model, data_init, num_t, min_t, flags):
    rig, seq = data_init["rigids_t"], data_init["seq_idx"]
    return (tuple(rig.shape), seq.detach().cpu().numpy().tobytes(), int(num_t), float(min_t), data_init.get("aatype") is None) + tuple(flags)


def inference_fn(model, diffuser, data_init, num_t, min_t, center=True, aux_traj=False, self_condition=True,
                 noise_scale=1.0, embed_self_conditioning=True, inpainting=False, input_aatype=False, noise_tape=None,
                 return_device=False, streams=1, experimental_streams=False, graph=False):
    """Same arguments / returned keys as the reference.  ``noise_tape=(z_rot, z_trans)`` ([num_t-1,B,N,3] float64
    N(0,1) draws) overrides the global ``np.random`` stream (sample-sharded runs).  ``data_init`` tensors carry a
    leading batch dimension B >= 1 (the reference always passes B = 1).  ``streams=n``: the batch runs as n sub-batches on n HIP
    streams (same results; the latency-bound node path of one sub-batch overlaps the pair kernels of the other) — experimental: needs
    ``experimental_streams=True`` (or FDIPT_EXPERIMENTAL_STREAMS=1), at most two streams, N <= 384 (``StreamedLoops``).
    ``graph=True``: the trajectory of this shape is captured once as a HIP graph and replayed for later batches (``GraphedTrajectory``)."""
    if graph:
        # whole-trajectory HIP graph, cached per (model, shape, seq_idx, schedule, options): first call of a shape eager (it also runs every
        # kernel once, which capture needs), second call captures, later calls replay
        if streams > 1:
            raise ValueError("graph=True replays one stream's launches: not combined with streams > 1")
        if noise_tape is None:
            n_noisy = int(np.sum(np.linspace(min_t, 1.0, num_t)[::-1] > min_t))
            noise_tape = draw_noise_tape(diffuser, n_noisy, data_init["rigids_t"].shape[0], data_init["rigids_t"].shape[1])
        cache = model.__dict__.setdefault("_graphed_trajectories", {})
        key = _graph_key(model, data_init, num_t, min_t, (center, aux_traj, self_condition, float(noise_scale), embed_self_conditioning,
                                                            inpainting, input_aatype, id(diffuser)))
        if key in cache:
            if cache[key] is None:
                cache[key] = GraphedTrajectory(model, diffuser, data_init, num_t, min_t, noise_tape, center=center, aux_traj=aux_traj,
                                               self_condition=self_condition, noise_scale=noise_scale,
                                               embed_self_conditioning=embed_self_conditioning, inpainting=inpainting, input_aatype=input_aatype)
                return cache[key].run(return_device=return_device)  # (captured on this very batch: its buffers hold it already)
            return cache[key].run(data_init, noise_tape, return_device)
        cache[key] = None  # (seen once: the next call of this shape captures)
    if streams > 1 and data_init["rigids_t"].shape[0] > 1:  # sub-batches on their own HIP streams (same results)
        loop = StreamedLoops(model, diffuser, data_init, streams, num_t, min_t, noise_tape=noise_tape, center=center, aux_traj=aux_traj,
                             self_condition=self_condition, noise_scale=noise_scale, embed_self_conditioning=embed_self_conditioning,
                             inpainting=inpainting, input_aatype=input_aatype, experimental=experimental_streams)
    else:
        loop = ReverseLoop(model, diffuser, data_init, num_t, min_t, center, aux_traj, self_condition, noise_scale,
                           embed_self_conditioning, inpainting, input_aatype, noise_tape)
    loop.prime()
    for k in range(num_t):
        loop.step(k)
    return loop.results(return_device)
