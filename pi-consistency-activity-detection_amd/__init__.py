"""MI355X-native hot path of AKASH2907/pi-consistency-activity-detection.

The directory name carries a hyphen (it mirrors the reference repo's name), so the package is
also registered under the importable alias ``picons_amd`` (see /picons_amd.py at the repo root).
Only what the semi-supervised train step needs lives here: csrc/ (HIP kernels + C-ABI),
the ctypes loader, the op-plan builder/executor and the host-side mirror of the reference's
module surface (dropin/).
"""
import importlib
import sys

_ALIAS = "picons_amd"
_me = sys.modules[__name__]
sys.modules.setdefault(_ALIAS, _me)

_SUBMODULES = ["spec", "synthetic", "capi", "desc", "ops", "spectral", "tail6", "plan", "model", "step", "dist"]


def _load_submodules():
    for sub in _SUBMODULES:
        try:
            m = importlib.import_module(__name__ + "." + sub)
        except ModuleNotFoundError as e:  # submodule not written yet (early rounds)
            if e.name and e.name.endswith("." + sub):
                continue
            raise
        setattr(_me, sub, m)
        sys.modules[_ALIAS + "." + sub] = m
        sys.modules[__name__ + "." + sub] = m


_load_submodules()
