"""The reference's registry / config entry points keep working with this repository on the path (BASELINE.json
north_star: "... the detectron2 META_ARCH_REGISTRY/config entry points so projects/UNINEXT configs load unchanged").

TEST-SIDE PROOF ONLY: skipped where /root/reference is absent (the GPU box), and nothing here is product code.
The build image lacks fvcore / yacs / iopath / omegaconf / termcolor, which `detectron2.config` and the registries
import, so this file installs THROW-AWAY stand-ins for exactly those third-party modules (a yacs-compatible CfgNode
on PyYAML, fvcore's Registry, iopath's PathManager) for the duration of the tests and removes them afterwards.
Everything else is the reference's own code, executed from where it lies:

  1. `detectron2.config.get_cfg()` + `uninext/config.py::add_uninext_config` + `merge_from_file` load
     projects/UNINEXT/configs/image_joint_r50.yaml and obj365v2_32g_r50.yaml unchanged;
  2. `detectron2/modeling/meta_arch/build.py`'s META_ARCH_REGISTRY resolves "UNINEXT_IMG" -- the name the yaml sets --
     to the reference's class (cut out of uninext_img.py with `ast`: the file imports cv2, skimage, the whole model zoo);
  3. the `DeformableTransformerEncoderLayer` the reference's transformer builds (deformable_transformer_dino.py:330-370,
     self_attn = MSDeformAttn(...) at :338) runs, with the reference's own ops/modules/ms_deform_attn.py and
     ops/functions/ms_deform_attn_func.py, on this repository's `MultiScaleDeformableAttention` module -- i.e. on the
     C ABI of libmsda_hip.so (host-pointer variants here; the same two functions drive the HIP kernels on a GPU) --
     and agrees with the mirror in uninext_amd.modules that loads the same state_dict.
"""
import ast
import copy
import importlib
import importlib.util
import os
import sys
import types
import warnings

import pytest
import torch
import yaml

REF = "/root/reference"
UX = os.path.join(REF, "projects/UNINEXT/uninext")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "detectron2")), reason="reference checkout not present")


# ---- throw-away stand-ins for the missing third-party modules -------------------------------------------------------
class _CfgNode(dict):
    """The subset of yacs.config.CfgNode (+ fvcore's load_yaml_with_base) that detectron2.config uses."""
    IMMUTABLE, NEW_ALLOWED = "__immutable__", "__new_allowed__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        self.__dict__[_CfgNode.IMMUTABLE] = False
        self.__dict__[_CfgNode.NEW_ALLOWED] = new_allowed
        for k, v in (init_dict or {}).items():
            self[k] = type(self)(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def is_frozen(self):
        return self.__dict__[_CfgNode.IMMUTABLE]

    def _set_frozen(self, flag):
        self.__dict__[_CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, _CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return v
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @classmethod
    def load_yaml_with_base(cls, filename, allow_unsafe=False):
        with open(filename) as f:
            cfg = yaml.safe_load(f)
        base = cfg.pop("_BASE_", None)
        if base is not None:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = cls.load_yaml_with_base(base, allow_unsafe)

            def rec(a, b):
                for k, v in a.items():
                    if isinstance(v, dict) and isinstance(b.get(k), dict):
                        rec(v, b[k])
                    else:
                        b[k] = v
            rec(cfg, merged)
            return merged
        return cfg

    def merge_from_other_cfg(self, other):
        def rec(a, b, path):
            for k, v in a.items():
                v = _CfgNode._decode(v)
                if k not in b:
                    if b.__dict__[_CfgNode.NEW_ALLOWED]:
                        b[k] = v
                        continue
                    raise KeyError("Non-existent config key: {}".format(".".join(path + [k])))
                if isinstance(b[k], _CfgNode):
                    rec(v, b[k], path + [k])
                else:
                    old = b[k]
                    if isinstance(old, tuple) and isinstance(v, list):
                        v = tuple(v)
                    elif isinstance(old, list) and isinstance(v, tuple):
                        v = list(v)
                    elif old is not None and v is not None and type(old) is not type(v) and not (
                            isinstance(old, (int, float)) and isinstance(v, (int, float))):
                        raise ValueError("Type mismatch for {}: {} vs {}".format(".".join(path + [k]), type(old), type(v)))
                    b[k] = v
        rec(other, self, [])

    def merge_from_list(self, cfg_list):
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            keys = full_key.split(".")
            for k in keys[:-1]:
                d = d[k]
            d[keys[-1]] = _CfgNode._decode(v)


class _Registry:
    def __init__(self, name):
        self._name, self._obj_map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._obj_map[o.__name__] = o
                return o
            return deco
        self._obj_map[obj.__name__] = obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


class _PathManager:
    def open(self, path, mode="r", **kw):
        return open(path, mode)

    def isfile(self, path):
        return os.path.isfile(path)

    def get_local_path(self, path, **kw):
        return path

    def register_handler(self, handler, **kw):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


@pytest.fixture(scope="module")
def d2():
    """sys.modules / sys.path with the stand-ins and the reference on them; restored afterwards."""
    saved_modules, saved_path = dict(sys.modules), list(sys.path)
    for k in [k for k in sys.modules if k.split(".")[0] in ("detectron2", "fvcore", "iopath", "omegaconf", "termcolor", "refops")]:
        del sys.modules[k]
    ph = type("PathHandler", (), {})
    stubs = {
        "fvcore": _mod("fvcore", __version__="0.1.5"), "fvcore.common": _mod("fvcore.common"),
        "fvcore.common.config": _mod("fvcore.common.config", CfgNode=_CfgNode),
        "fvcore.common.registry": _mod("fvcore.common.registry", Registry=_Registry),
        "iopath": _mod("iopath"), "iopath.common": _mod("iopath.common"),
        "iopath.common.file_io": _mod("iopath.common.file_io", PathManager=_PathManager, PathHandler=ph,
                                      HTTPURLHandler=type("HTTPURLHandler", (ph,), {}),
                                      OneDrivePathHandler=type("OneDrivePathHandler", (ph,), {})),
        "omegaconf": _mod("omegaconf", DictConfig=type("DictConfig", (), {}), ListConfig=type("ListConfig", (), {}),
                          OmegaConf=type("OmegaConf", (), {}), SCMode=type("SCMode", (), {})),
        "termcolor": _mod("termcolor", colored=lambda s, *a, **k: s),
    }
    sys.modules.update(stubs)
    sys.path.insert(0, REF)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yield importlib.import_module("detectron2.config")
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_modules:
                del sys.modules[k]
        sys.modules.update(saved_modules)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("yaml_name", ["image_joint_r50.yaml", "obj365v2_32g_r50.yaml"])
def test_projects_uninext_configs_load_unchanged(d2, yaml_name):
    assert d2.__file__.startswith(REF)                                       # the reference's own detectron2.config
    add_uninext_config = _load(os.path.join(UX, "config.py"), "ref_uninext_config").add_uninext_config
    cfg = d2.get_cfg()
    add_uninext_config(cfg)
    cfg.merge_from_file(os.path.join(REF, "projects/UNINEXT/configs", yaml_name))
    cfg.freeze()
    assert cfg.MODEL.META_ARCHITECTURE == "UNINEXT_IMG"
    assert cfg.MODEL.DDETRS.NUM_OBJECT_QUERIES == 900 and cfg.MODEL.OTA is True
    # the geometry the HIP path is specialised for (d_model 256, 8 heads, 4 levels, 4 points, 6 + 6 layers)
    dd = cfg.MODEL.DDETRS
    assert (dd.HIDDEN_DIM, dd.NHEADS, dd.NUM_FEATURE_LEVELS, dd.ENC_N_POINTS, dd.DEC_N_POINTS, dd.ENC_LAYERS, dd.DEC_LAYERS) == \
        (256, 8, 4, 4, 4, 6, 6)
    with pytest.raises(AttributeError):
        cfg.MODEL.OTA = False                                                    # frozen, like any detectron2 config


def test_meta_arch_registry_resolves_uninext_img(d2):
    build = _load(os.path.join(REF, "detectron2/modeling/meta_arch/build.py"), "ref_meta_arch_build")
    reg = build.META_ARCH_REGISTRY
    src = open(os.path.join(UX, "uninext_img.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "UNINEXT_IMG"]
    assert len(cls) == 1 and any("META_ARCH_REGISTRY" in ast.dump(dec) for dec in cls[0].decorator_list)
    ns = {"META_ARCH_REGISTRY": reg, "nn": torch.nn, "torch": torch}
    exec(compile(ast.Module(body=cls, type_ignores=[]), os.path.join(UX, "uninext_img.py"), "exec"), ns)
    add_uninext_config = _load(os.path.join(UX, "config.py"), "ref_uninext_config").add_uninext_config
    cfg = d2.get_cfg()
    add_uninext_config(cfg)
    cfg.merge_from_file(os.path.join(REF, "projects/UNINEXT/configs/image_joint_r50.yaml"))
    got = reg.get(cfg.MODEL.META_ARCHITECTURE)                                   # what build_model(cfg) looks up (build.py:20-21)
    assert got is ns["UNINEXT_IMG"] and issubclass(got, torch.nn.Module)
    with pytest.raises(KeyError):
        reg.get("NOT_A_MODEL")


def test_transformer_layer_runs_on_this_repositorys_module(d2):
    """deformable_transformer_dino.py:330-370 with the reference's own MSDeformAttn on top of ./MultiScaleDeformableAttention.py."""
    ops = os.path.join(UX, "models/deformable_detr/ops")
    pkg = types.ModuleType("refops"); pkg.__path__ = [ops]
    sys.modules["refops"] = pkg
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for sub in ("functions", "modules"):
            spec = importlib.util.spec_from_file_location("refops." + sub, os.path.join(ops, sub, "__init__.py"),
                                                          submodule_search_locations=[os.path.join(ops, sub)])
            mod = importlib.util.module_from_spec(spec)
            sys.modules["refops." + sub] = mod
            spec.loader.exec_module(mod)
    func = sys.modules["refops.functions.ms_deform_attn_func"]
    from uninext_amd import ext
    assert func.MSDA.ms_deform_attn_forward is ext.ms_deform_attn_forward        # the drop-in boundary

    dino = os.path.join(UX, "models/deformable_detr/deformable_transformer_dino.py")
    body = [n for n in ast.parse(open(dino).read()).body
            if (isinstance(n, ast.ClassDef) and n.name == "DeformableTransformerEncoderLayer")
            or (isinstance(n, ast.FunctionDef) and n.name == "_get_activation_fn")]
    ns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "MSDeformAttn": sys.modules["refops.modules"].MSDeformAttn}
    exec(compile(ast.Module(body=body, type_ignores=[]), dino, "exec"), ns)
    theirs = ns["DeformableTransformerEncoderLayer"](d_model=256, d_ffn=512, dropout=0.0, n_levels=4, n_heads=8, n_points=4).eval()
    assert type(theirs.self_attn).__module__ == "refops.modules.ms_deform_attn"   # the reference's class, built at :338
    with torch.no_grad():
        theirs.self_attn.sampling_offsets.weight.normal_(0, 0.02)
        theirs.self_attn.attention_weights.weight.normal_(0, 0.05)

    from uninext_amd import workloads
    from uninext_amd.modules import DeformableTransformerEncoderLayer
    ours = DeformableTransformerEncoderLayer(d_model=256, d_ffn=512, dropout=0.0, n_heads=8).eval()
    ours.load_state_dict(theirs.state_dict())                                     # same parameter names
    levels = ((9, 12), (5, 6), (3, 3), (2, 2))
    S = sum(h * w for h, w in levels)
    shapes, lsi = workloads.level_tensors(levels, "cpu")
    g = torch.Generator().manual_seed(4)
    src, pos = torch.randn(2, S, 256, generator=g), torch.randn(2, S, 256, generator=g) * 0.3
    ref = workloads.encoder_reference_points(levels, "cpu")[None, :, None, :].expand(2, S, 4, 2).contiguous()
    mask = torch.zeros(2, S, dtype=torch.bool)
    mask[1, -9:] = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = theirs(src, pos, ref, shapes, lsi, mask)
    b = ours(src, pos, ref, shapes, lsi, mask)
    assert (a - b).abs().max().item() < 1e-5
