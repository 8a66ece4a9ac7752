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


def import_reference():
    """Returns the reference's `src.d_fine` package modules as a namespace."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
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
    return ns
