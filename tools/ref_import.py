"""Import the reference (/root/reference) on CPU in the AUTHORING container only.

Never shipped to / used on the GPU box.  Shims:
  1. sys.modules stubs for third-party imports that are not installed here
     (torchsummary, torchvision, tensorboardX, imageio, skvideo.io, cv2, wandb, matplotlib ok);
  2. `datasets`, `models`, `utils` pre-registered as namespace packages rooted at the
     reference (HF `datasets` in site-packages would otherwise shadow it);
  3. torch.cuda.FloatTensor -> torch.FloatTensor, Tensor.cuda / Module.cuda -> identity;
  4. a synthetic `rgb_charades.pt` (the real one is a download) written to a temp dir.
"""
import importlib.machinery
import os
import sys
import tempfile
import types

import torch
import torch.nn as nn

REF = os.environ.get("PICONS_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims(double=False):
    if not os.path.isdir(REF):
        raise RuntimeError("reference not present at %s (authoring container only)" % REF)
    for n in ("torchsummary", "tensorboardX", "imageio", "cv2", "wandb"):
        if n not in sys.modules:
            _stub(n, summary=lambda *a, **k: None, SummaryWriter=object)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.datasets = _stub("torchvision.datasets")
        tv.transforms = _stub("torchvision.transforms")
    if "skvideo" not in sys.modules:
        sk = _stub("skvideo")
        sk.io = _stub("skvideo.io", vread=lambda *a, **k: None)
    for pkg in ("datasets", "models", "utils"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        m.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=True)
        m.__spec__.submodule_search_locations = m.__path__
        sys.modules[pkg] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.cuda.FloatTensor = torch.DoubleTensor if double else torch.FloatTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def synthetic_charades(state, path=None):
    """Write the trunk part of a synthetic state (keys without the 'conv1.' prefix) as the
    `rgb_charades.pt` CapsNet.__init__ loads (capsules_ucf101.py:343-352)."""
    path = path or os.path.join(tempfile.mkdtemp(prefix="picons_ref_"), "rgb_charades.pt")
    sd = {k[len("conv1."):]: torch.from_numpy(v.copy()) if hasattr(v, "shape") else v
          for k, v in state.items() if k.startswith("conv1.")}
    torch.save(sd, path)
    return path


class ScriptedDropout(nn.Module):
    """Stands in for the reference's shared nn.Dropout3d (capsules_ucf101.py:371) so the draws
    are the scripted per-(sample,channel) scales instead of torch RNG."""

    def __init__(self, scales):
        super().__init__()
        self.scales = list(scales)
        self.i = 0

    def forward(self, x):
        s = self.scales[self.i]
        self.i += 1
        if s is None:
            return x
        return x * torch.as_tensor(s).to(x.dtype).view(x.shape[0], -1, 1, 1, 1)
