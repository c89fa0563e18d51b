"""ctypes binding of ``libfdipt_hip.so`` (C ABI in ``include/fdipt.h``).

The product path has no CPU fallback: every arithmetic entry point goes through this
library and raises if it is missing or if a kernel reports an error.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDIPT_LIB: another build of the library (the -DFDIPT_DEV build that tools/ use: csrc/build.sh with FDIPT_DEV=1)
LIB_PATH = os.environ.get("FDIPT_LIB") or os.path.join(_HERE, "lib", "libfdipt_hip.so")

PREC_F32, PREC_BF16, PREC_F16 = 0, 1, 2
# FdiptDims.kernel_flags (include/fdipt.h): fallback paths of the half-precision mode, for parity tests
KF_ET3, KF_GENERIC_PAIR, KF_GENERIC_ATTN, KF_UNFUSED_NODE, KF_UNFOLDED, KF_NO_SPLIT, KF_NO_MERGE, KF_ROWS32, KF_PASS_Z = 1, 2, 4, 8, 16, 32, 64, 128, 256
_ERR = {-1: "FDIPT_EINVAL (bad argument)", -2: "FDIPT_ELAUNCH (HIP launch error)",
        -3: "FDIPT_ESIZE (workspace too small or N beyond the compiled tiling)"}


class FdiptError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "c_s", "c_z", "c_hidden", "c_skip", "no_heads", "no_qk_points", "no_v_points", "tfmr_heads", "tfmr_layers",
        "num_blocks", "index_embed", "num_bins", "use_aatype", "precision", "kernel_flags")] + [
        (n, C.c_float) for n in ("min_bin", "max_bin", "coordinate_scaling", "r3_min_b", "r3_max_b")]


_P = C.c_void_p


class ForwardArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n_rel", C.c_int32), ("rel_off", C.c_int32)] + [
        (n, _P) for n in ("rigids_t", "res_mask", "fixed_mask", "sc_ca_t", "seq_idx", "idx_emb", "aatype", "gt_psi", "t",
                          "t_emb", "t_emb_eps", "so3_sigma", "bb_tables", "psi", "rot_score", "trans_score", "rigids",
                          "atom37", "atom14", "trace_node", "trace_edge", "trace_inner")] + [
        ("ev_start", C.POINTER(C.c_void_p)), ("ev_stop", C.POINTER(C.c_void_p)), ("ca_out", _P), ("reserve_cus", C.c_int32), ("clock_out", _P),
                ("so3_score_table", _P), ("so3_omega_edges", _P), ("so3_num_omega", C.c_int32), ("step_cursor", _P)]


class ReverseIndexed(C.Structure):
    """FdiptReverseIndexed (include/fdipt.h): one reverse step addressed through a device-side step cursor."""
    _fields_ = [("B", C.c_int32), ("N", C.c_int32)] + [(n, _P) for n in (
        "rigid_traj", "rot_score", "trans_score", "diffuse_mask", "z_rot", "z_trans", "t_table")] + [
        ("dt", C.c_double), ("noise_scale", C.c_double), ("center", C.c_int32), ("diffuse_rot", C.c_int32), ("diffuse_trans", C.c_int32)] + [
        (n, C.c_double) for n in ("so3_min_sigma", "so3_max_sigma", "r3_min_b", "r3_max_b", "coordinate_scaling")] + [
        (n, _P) for n in ("psi", "aatype", "bb_tables", "prot_traj", "pred_rigids", "traj_fixed_mask", "trans_traj", "step_cursor")]


_lib = None

# name -> (restype, argtypes); every symbol declared in include/fdipt.h
_i, _d, _f, _sz, _i64 = C.c_int, C.c_double, C.c_float, C.c_size_t, C.c_int64
_DP = C.POINTER(Dims)
SIGNATURES = {
    "fdipt_param_count": (_i, [_DP]),
    "fdipt_param_offset": (_i64, [_DP, _i]),
    "fdipt_derived_bytes": (_sz, [_DP]),
    "fdipt_model_prepare": (_i, [_DP, _P, _P, _P]),
    "fdipt_setup_bytes": (_sz, [_DP, _i, _i, _i]),
    "fdipt_sample_setup": (_i, [_DP, _P, _P, _i, _i, _i, _P, _P, _P]),
    "fdipt_forward_workspace_bytes": (_sz, [_DP, _i, _i]),
    "fdipt_score_forward": (_i, [_DP, _P, _P, _P, C.POINTER(ForwardArgs), _P, _sz, _P]),
    "fdipt_edge_embed_fwd": (_i, [_DP, _P, _P, _P, C.POINTER(ForwardArgs), _P, _P, _P, _sz, _P]),
    "fdipt_ipa_project_points": (_i, [_DP, _P, _P, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "fdipt_ipa_attention_fwd": (_i, [_DP, _P, _P, _i, _i, _i, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "fdipt_edge_transition_fwd": (_i, [_DP, _P, _P, _i, _i, _i, _P, _P, _P, _P, _P, _sz, _P]),
    "fdipt_se3_reverse_step": (_i, [_i, _i, _P, _P, _P, _P, _P, _P, _d, _d, _d, _i, _i, _i, _d, _d, _d, _d, _d, _P, _P, _P]),
    "fdipt_se3_reverse_step_atoms": (_i, [_i, _i, _P, _P, _P, _P, _P, _P, _d, _d, _d, _i, _i, _i, _d, _d, _d, _d, _d, _P, _P,
                                          _P, _P, _P, _P, _P]),
    "fdipt_se3_reverse_step_traj": (_i, [_i, _i, _P, _P, _P, _P, _P, _P, _d, _d, _d, _i, _i, _i, _d, _d, _d, _d, _d, _P, _P,
                                         _P, _P, _P, _P, _P, _P, _P, _P]),
    "fdipt_se3_reverse_step_indexed": (_i, [C.POINTER(ReverseIndexed), _P]),
    "fdipt_backbone_atoms_indexed": (_i, [_i, _P, _P, _P, _P, _P, _P, _P]),
    "fdipt_se3_forward_step": (_i, [_i, _i, _P, _P, _P, _P, _P, _d, _d, _d, _d, _d, _d, _d, _d, _P, _P, _P, _P]),
    "fdipt_se3_step_log_prob": (_i, [_i, _i, _P, _P, _P, _P, _P, _P, _P, _d, _d, _d, _d, _d, _d, _d, _d, _P, _P]),
    "fdipt_se3_prior_log_prob": (_i, [_i, _i, _P, _P, _d, _P, _P]),
    "fdipt_quat_to_rot": (_i, [_i, _P, _P, _P]),
    "fdipt_rot_to_quat": (_i, [_i, _P, _P, _P]),
    "fdipt_quat_multiply": (_i, [_i, _P, _P, _P, _P]),
    "fdipt_quat_multiply_by_vec": (_i, [_i, _P, _P, _P, _P]),
    "fdipt_invert_quat": (_i, [_i, _P, _P, _P]),
    "fdipt_rigid_apply": (_i, [_i, _P, _P, _P, _P]),
    "fdipt_rigid_invert_apply": (_i, [_i, _P, _P, _P, _P]),
    "fdipt_rigid_compose": (_i, [_i, _P, _P, _P, _P, _P]),
    "fdipt_rigid_invert": (_i, [_i, _P, _P, _P, _P]),
    "fdipt_rigid_compose_q_update": (_i, [_i, _P, _P, _P, _P, _P]),
    "fdipt_quat_to_rotvec": (_i, [_i, _P, _P, _P]),
    "fdipt_rigid_from_3_points": (_i, [_i, _P, _P, _P, _f, _P, _P]),
    "fdipt_so3_exp_geomstats": (_i, [_i, _P, _P, _P]),
    "fdipt_so3_log_geomstats": (_i, [_i, _P, _P, _P]),
    "fdipt_so3_omega": (_i, [_i, _P, _d, _P, _P]),
    "fdipt_so3_exp": (_i, [_i, _P, _P, _P]),
    "fdipt_so3_log": (_i, [_i, _P, _P, _P]),
    "fdipt_igso3_rot_score": (_i, [_i, _i, _P, _P, _P, _P, _P, _P]),
    "fdipt_igso3_rot_score_cached": (_i, [_i, _i, _P, _P, _P, _P, _i, _P, _P, _P]),
    "fdipt_r3_trans_score": (_i, [_i, _i, _P, _P, _P, _f, _f, _f, _P, _P, _P]),
    "fdipt_backbone_atoms": (_i, [_i, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fdipt_linear": (_i, [_i, _i, _i, _i, _P, _i, _P, _i, _P, _P, _i, _P, _i, _P, _i, _P]),
    "fdipt_layernorm": (_i, [_i, _i, _P, _P, _P, _P, _P, _P, _P]),
    "fdipt_selftest_mfma": (_i, [_i, C.POINTER(_d)]),
    "fdipt_event_create": (_i, [C.POINTER(_P)]),
    "fdipt_event_destroy": (_i, [_P]),
    "fdipt_event_record": (_i, [_P, _P]),
    "fdipt_event_elapsed_ms": (_i, [_P, _P, C.POINTER(_f)]),
    "fdipt_version": (C.c_char_p, []),
    "fdipt_kernel_class_bounds": (_i, [C.POINTER(C.c_int32), _i]),
}


def load():
    """Load the shared library (once) and bind every declared symbol; raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FdiptError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or framedipt_amd/csrc/build.sh).  There is no CPU fallback for the sampler hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise FdiptError(f"libfdipt_hip: {what} failed with {_ERR.get(rc, rc)}")


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor / None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise FdiptError("non-contiguous tensor passed to the C ABI")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, what: str):
    if not t.is_cuda:
        raise FdiptError(f"{what}: tensors must live on the MI355X (got device {t.device}); no CPU fallback exists")
