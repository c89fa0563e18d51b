"""Checkpoint ingest: the reference's ``.pth`` / ``.pkl`` checkpoints -> weights + configuration.

A reference checkpoint (``framedipt/data/utils.py:381-417 write_checkpoint``) is ``torch.save`` of
``{"model": state_dict, "conf": omegaconf.DictConfig, "optim": ..., "epoch": ..., "step": ...}``; the state-dict keys may carry a
``module.`` prefix (DataParallel), and ``Inference._load_ckpt`` (``experiments/inference.py:107-161``) merges ``conf.model`` over
the run's model config and takes ``conf.diffuser.r3`` wholesale before it builds ``SE3Diffuser`` / ``ScoreNetwork``.

Unpickling the ``DictConfig`` needs the ``omegaconf`` package, which this image does not have: ``load_checkpoint`` unpickles with
a stand-in for every ``omegaconf.*`` class that only keeps the pickled state, and then reads the configuration out of that state
(omegaconf 2.x layout: containers hold ``_content`` - a dict / list of nodes -, value nodes hold ``_val``).  ``${a.b}``
interpolations that point inside the same configuration are resolved; anything else stays a string.

    python -m framedipt_amd.checkpoint weights.pth out_prefix      # -> out_prefix.npz (state dict) + out_prefix.yaml (conf)
"""
from __future__ import annotations

import pickle
import re
import sys
import types
from collections import OrderedDict

import numpy as np


class _Stub:
    """Stand-in for an omegaconf class: keeps whatever state the pickle carries."""

    def __init__(self, *args, **kwargs):
        self.__dict__["_ctor_args"] = (args, kwargs)

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):  # (dict state, slots state)
            state = {**(state[0] or {}), **state[1]}
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    def __setattr__(self, k, v):
        self.__dict__[k] = v

    def __reduce_ex__(self, protocol):  # pragma: no cover - stubs are never re-pickled
        raise pickle.PicklingError("omegaconf stand-ins are read-only")


_stub_classes: dict = {}


def _stub_for(module: str, name: str):
    key = (module, name)
    if key not in _stub_classes:
        _stub_classes[key] = type(name, (_Stub,), {"__module__": module, "_fd_stub": True})
    return _stub_classes[key]


# Globals a FrameDiPT checkpoint may name besides the omegaconf classes: the tensor-rebuild helpers of torch.save, the containers
# of a state dict / optimizer state, the enums / typing objects an omegaconf 2.x container pickles.  Everything else is refused:
# unpickling executes the callables a pickle names, so an untrusted ``.pth`` must not get to choose them.  Deliberately absent:
# ``builtins.getattr`` / ``builtins.object`` (with a rebuild helper in reach, ``getattr(f, "__globals__")`` walks to ``sys.modules`` and
# from there to ``os.system``), ``pathlib`` (its classes touch the file system), and any dotted name (protocol 4 resolves
# ``a.b.c`` attribute by attribute, i.e. it is ``getattr`` again).  ``omegaconf.*`` names ALWAYS become inert stand-ins that only keep
# the pickled state — also when the real package is installed: its classes are not needed to read the configuration out of the state.
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "set"),
    ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "complex"),
    ("builtins", "slice"), ("typing", "Any"), ("typing", "Union"), ("typing", "Optional"),
    ("typing", "Dict"), ("typing", "List"), ("typing", "Tuple"),
    ("enum", "Enum"), ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"), ("_codecs", "encode"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Size"), ("torch", "device"),
    ("torch.serialization", "_get_layout"), ("torch.nn.parameter", "Parameter"),
}
_ALLOWED_PREFIX = (("torch", "Storage"), ("torch", "Tensor"))  # torch.FloatStorage ..., torch.FloatTensor ... (legacy type names)
_TORCH_DTYPES = {"float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool", "float", "double",
                 "half", "long", "int", "short"}


class _ShimUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if "." in name or not name.isidentifier():
            raise pickle.UnpicklingError(f"checkpoint names the dotted global {module}.{name}: refused (attribute walks are getattr in disguise; "
                                         "see framedipt_amd/checkpoint.py:_ALLOWED_GLOBALS)")
        if module == "omegaconf" or module.startswith("omegaconf."):
            if not all(p.isidentifier() for p in module.split(".")):
                raise pickle.UnpicklingError(f"checkpoint names the global {module}.{name}: refused")
            return _stub_for(module, name)  # never the real class: only the pickled state is read (to_plain)
        if module == "__builtin__":  # protocol-2 spelling of builtins
            module = "builtins"
        if (module, name) in _ALLOWED_GLOBALS or (module == "torch" and (name in _TORCH_DTYPES or name.endswith(("Storage", "Tensor")))):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint names the global {module}.{name}, which a FrameDiPT checkpoint has no use for: refused "
                                     "(unpickling runs the callables a file names; see framedipt_amd/checkpoint.py:_ALLOWED_GLOBALS)")


# torch.load(pickle_module=...) wants a module-like object with Unpickler / load / loads
_shim = types.ModuleType("framedipt_amd._ckpt_pickle")
_shim.Unpickler = _ShimUnpickler
_shim.Pickler = pickle.Pickler
_shim.load = lambda f, **kw: _ShimUnpickler(f, **kw).load()
_shim.loads = pickle.loads
_shim.dump, _shim.dumps = pickle.dump, pickle.dumps
_shim.__name__ = "pickle"


def to_plain(node):
    """omegaconf container / node (real or stand-in) -> dict / list / scalar."""
    if isinstance(node, (str, int, float, bool, type(None))):
        return node
    if isinstance(node, dict):
        return {k: to_plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [to_plain(v) for v in node]
    d = getattr(node, "__dict__", {})
    if "_content" in d:
        return to_plain(d["_content"])
    if "_val" in d:
        return to_plain(d["_val"])
    if hasattr(node, "value") and type(node).__module__.startswith("enum"):  # pragma: no cover
        return node.value
    try:  # a real omegaconf object
        from omegaconf import OmegaConf  # type: ignore
        return OmegaConf.to_container(node, resolve=False)
    except Exception as e:  # noqa: BLE001
        raise TypeError(f"cannot read configuration node of type {type(node)}") from e


_INTERP = re.compile(r"^\$\{([A-Za-z0-9_.]+)\}$")


def resolve_interpolations(conf: dict) -> dict:
    """``${a.b.c}`` -> the value at that path of the same configuration (as OmegaConf resolves the three interpolations of
    config/base.yaml); unresolvable ones stay as they are."""
    def lookup(path):
        cur = conf
        for p in path.split("."):
            cur = cur[p]
        return cur

    def walk(x, depth=0):
        if isinstance(x, dict):
            return {k: walk(v, depth) for k, v in x.items()}
        if isinstance(x, list):
            return [walk(v, depth) for v in x]
        if isinstance(x, str):
            m = _INTERP.match(x)
            if m and depth < 8:
                try:
                    return walk(lookup(m.group(1)), depth + 1)
                except (KeyError, TypeError):
                    return x
        return x
    return walk(conf)


def load_checkpoint(path, map_location="cpu"):
    """-> (state_dict: OrderedDict name -> float32 ndarray without the ``module.`` prefix, conf: plain nested dict or None,
    extras: {"epoch", "step"} when present)."""
    import torch
    try:
        # (weights_only=False: the omegaconf container is not a tensor; what may be constructed is restricted by _ShimUnpickler instead)
        ckpt = torch.load(path, map_location=map_location, pickle_module=_shim, weights_only=False)
    except pickle.UnpicklingError as e:
        if "refused" in str(e):
            raise
        with open(path, "rb") as f:  # plain pickle written with use_torch=False
            ckpt = _ShimUnpickler(f).load()
    except (RuntimeError, EOFError):  # plain pickle written with use_torch=False
        with open(path, "rb") as f:
            ckpt = _ShimUnpickler(f).load()
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise ValueError(f"{path}: not a FrameDiPT checkpoint (expected a dict with a 'model' entry)")
    sd = OrderedDict()
    for k, v in ckpt["model"].items():
        arr = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        sd[k.replace("module.", "")] = arr.astype(np.float32, copy=False)  # inference.py:156-158
    conf = ckpt.get("conf")
    conf = resolve_interpolations(to_plain(conf)) if conf is not None else None
    return sd, conf, {k: ckpt[k] for k in ("epoch", "step") if k in ckpt}


def merge(base: dict, over: dict) -> dict:
    """OmegaConf.merge for plain dicts: ``over`` wins, dictionaries merge recursively."""
    out = dict(base)
    for k, v in over.items():
        out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def apply_checkpoint_conf(cfg, ckpt_conf: dict, seed=None, conf_overrides: dict | None = None):
    """The configuration steps of ``Inference._load_ckpt`` (inference.py:133-148) on a ``framedipt_amd.config.Conf``: model <-
    merge(model, ckpt.model); diffuser.r3 <- ckpt.diffuser.r3; both diffuser seeds <- the inference seed.  ``conf_overrides``
    (the caller's overrides, inference.py:117,146-147) is merged last and wins over the checkpoint's values."""
    from .config import to_conf
    cfg = to_conf(merge(dict(cfg), {"model": ckpt_conf.get("model", {})}))
    if "diffuser" in ckpt_conf and "r3" in ckpt_conf["diffuser"]:
        cfg.diffuser.r3 = to_conf(dict(ckpt_conf["diffuser"]["r3"]))
    if conf_overrides:  # (after the checkpoint's diffuser.r3, as the reference: an override of e.g. diffuser.r3.min_b wins)
        cfg = to_conf(merge(dict(cfg), conf_overrides))
    m = cfg.model
    # fields the library reads from the ipa / embed sub-configs (config/base.yaml interpolations)
    m.ipa.setdefault("c_s", m.node_embed_size)
    m.ipa.setdefault("c_z", m.edge_embed_size)
    m.ipa.setdefault("coordinate_scaling", cfg.diffuser.r3.coordinate_scaling)
    m.embed.min_bin = float(m.embed.min_bin)
    if seed is None:
        seed = cfg.get("inference", {}).get("seed", 123)
    cfg.diffuser.so3.seed = cfg.diffuser.r3.seed = seed
    cfg.diffuser.r3.setdefault("seed", seed)
    return cfg


def load_model(weights_path, cfg=None, inpainting: bool = False, precision: str = "fp16", device="cuda", conf_overrides=None):
    """``Inference._load_ckpt``: checkpoint -> (cfg, SE3Diffuser, ScoreNetwork on ``device``)."""
    from . import config
    from .diffusion import SE3Diffuser
    from .model import ScoreNetwork
    sd, ckpt_conf, _ = load_checkpoint(weights_path)
    cfg = apply_checkpoint_conf(cfg if cfg is not None else config.base_config(inpainting), ckpt_conf or {}, conf_overrides=conf_overrides)
    diffuser = SE3Diffuser(cfg.diffuser, device=device)
    model = ScoreNetwork(cfg.model, diffuser, inpainting=inpainting, precision=precision)
    model.load_state_dict(sd)
    return cfg, diffuser, model.to(device).eval()


def main(argv=None):
    import yaml
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit(__doc__)
    sd, conf, extra = load_checkpoint(argv[0])
    np.savez(argv[1] + ".npz", **sd)
    with open(argv[1] + ".yaml", "w") as f:
        yaml.safe_dump({"conf": conf, **{k: int(v) for k, v in extra.items()}}, f, sort_keys=False)
    print(f"{len(sd)} tensors, {sum(v.size for v in sd.values())} parameters -> {argv[1]}.npz / .yaml")


if __name__ == "__main__":
    main()
