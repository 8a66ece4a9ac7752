"""Import the upstream reference (`/root/reference/src/d_fine`) inside THIS container only.

Used by the golden-vector generator (tools/gen_golden.py) and by ad-hoc parity probes.
Never imported by the product, the tests or the bench: `/root/reference` does not exist
on the GPU box.  Three import stubs are needed because the container lacks the packages
(SURVEY.md §8c): `loguru.logger`, `torchvision` and `torchvision.ops.boxes.box_area`.
"""
import sys
import types

REF_ROOT = "/root/reference"


def install_stubs():
    if "loguru" not in sys.modules:
        lg = types.ModuleType("loguru")

        class _L:
            def __getattr__(self, _name):
                return lambda *a, **k: None

        lg.logger = _L()
        sys.modules["loguru"] = lg
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        boxes = types.ModuleType("torchvision.ops.boxes")

        def box_area(b):
            return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

        boxes.box_area = box_area
        ops.boxes = boxes
        tv.ops = ops
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = ops
        sys.modules["torchvision.ops.boxes"] = boxes


def _bind_src_to_reference():
    """Makes `src` mean /root/reference/src.  The repo root carries its own `src` alias package (a regular
    package, so it would win over the reference's namespace package whenever the repo root is on sys.path, and
    its meta-path finder would route `src.d_fine.*` to THIS build): drop both and pin `src.__path__`."""
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.meta_path[:] = [f for f in sys.meta_path if type(f).__name__ != "_AliasFinder"]
    pkg = types.ModuleType("src")
    pkg.__path__ = [REF_ROOT + "/src"]
    sys.modules["src"] = pkg


def import_reference():
    """Returns the reference's `src.d_fine` package modules as a namespace."""
    install_stubs()
    _bind_src_to_reference()
    import importlib

    ns = types.SimpleNamespace()
    ns.dfine = importlib.import_module("src.d_fine.dfine")
    ns.configs = importlib.import_module("src.d_fine.configs")
    ns.matcher = importlib.import_module("src.d_fine.matcher")
    ns.criterion = importlib.import_module("src.d_fine.dfine_criterion")
    ns.arch_utils = importlib.import_module("src.d_fine.arch.utils")
    ns.decoder = importlib.import_module("src.d_fine.arch.dfine_decoder")
    ns.encoder = importlib.import_module("src.d_fine.arch.hybrid_encoder")
    ns.backbone = importlib.import_module("src.d_fine.arch.hgnetv2")
    for m in vars(ns).values():
        assert m.__file__.startswith(REF_ROOT + "/"), f"{m.__name__} resolved to {m.__file__}, not the reference"
    return ns


_DL_ABSENT = ("hydra", "onnx", "onnxsim", "openvino", "tensorrt", "omegaconf", "onnxconverter_common", "wandb",
              "cv2", "albumentations", "albumentations.core", "albumentations.core.transforms_interface",
              "albumentations.pytorch", "faster_coco_eval", "faster_coco_eval.core", "matplotlib",
              "matplotlib.pyplot", "torchmetrics", "torchmetrics.detection", "torchmetrics.detection.mean_ap")


def import_reference_dl():
    """`src.dl.export` / `src.dl.utils` of the reference (golden generator only).  Their module-level
    imports of tool-chains this container lacks (hydra, onnx, TensorRT, OpenVINO, cv2, albumentations, wandb ...)
    are satisfied by inert placeholder modules: none of the functions the generator calls
    (`DFINEPostProcessor`, `process_boxes`) touches them."""
    from unittest import mock
    ns = import_reference()
    for name in _DL_ABSENT:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = mock.MagicMock(name=name)
                m.__path__ = []
                import importlib.machinery
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)      # torch._dynamo probes find_spec("onnx") when an optimizer is built
                sys.modules[name] = m
    tv = sys.modules["torchvision"]
    if not hasattr(tv.ops, "nms"):
        tv.ops.nms = None
        tv.ops.box_iou = None
    import importlib
    ns.dl_utils = importlib.import_module("src.dl.utils")
    ns.dl_export = importlib.import_module("src.dl.export")
    assert ns.dl_export.__file__.startswith(REF_ROOT + "/") and ns.dl_utils.__file__.startswith(REF_ROOT + "/")
    return ns
