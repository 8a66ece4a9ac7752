"""Alias package: makes the reference's import paths resolve to this build, so that code written
against `src.d_fine.*` / `src.dl.train` / `src.dl.export` runs on the MI355X-native implementation
unchanged.

    from src.d_fine.dfine import build_model, build_loss, build_optimizer
    from src.d_fine.matcher import HungarianMatcher
    python -m src.dl.train model_name=m train.batch_size=32
"""
import importlib
import sys

_ALIASES = {
    "src.d_fine": "custom_d_fine_amd.d_fine",
    "src.d_fine.arch": "custom_d_fine_amd.d_fine.arch",
    "src.dl": "custom_d_fine_amd.dl",
    "src.infer": "custom_d_fine_amd.infer",
    "src.infer.torch_model": "custom_d_fine_amd.infer.torch_model",
}
for _name in ("dfine", "configs", "matcher", "dfine_criterion", "dist_utils", "utils"):
    _ALIASES[f"src.d_fine.{_name}"] = f"custom_d_fine_amd.d_fine.{_name}"
for _name in ("hgnetv2", "common", "hybrid_encoder", "dfine_decoder", "utils"):
    _ALIASES[f"src.d_fine.arch.{_name}"] = f"custom_d_fine_amd.d_fine.arch.{_name}"
for _name in ("train", "export", "engine", "synthetic", "fused_optim", "postprocess", "validator"):
    _ALIASES[f"src.dl.{_name}"] = f"custom_d_fine_amd.dl.{_name}"


class _AliasFinder:
    """Resolves `src.*` lazily so importing `src` does not pull torch in."""

    @staticmethod
    def find_spec(name, path=None, target=None):
        real = _ALIASES.get(name)
        if real is None:
            return None
        from importlib.machinery import ModuleSpec

        class _Loader:
            @staticmethod
            def create_module(spec):
                return importlib.import_module(real)

            @staticmethod
            def exec_module(module):
                pass

        return ModuleSpec(name, _Loader(), is_package=real.count(".") < 3 and not real.split(".")[-1] in
                          ("dfine", "configs", "matcher", "dfine_criterion", "dist_utils", "utils", "train", "export", "torch_model"))


sys.meta_path.insert(0, _AliasFinder())
