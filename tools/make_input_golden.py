#!/usr/bin/env python3
"""AUTHORING CONTAINER ONLY: run the reference's own `UCF101DataLoader.__getitem__` / `load_video`
(/root/reference/datasets/ucf_dataloader.py) on the synthetic decoded videos of tests/inputfixture.py and record its
samples in tests/golden/input_pipe.npz.  Stubs: `skvideo.io.vread` returns the synthetic frames, `cv2.resize` is the
identity it is for a 224x224 crop resized to 224x224 (:156,:162; anything else raises), the annotation pickles are not read
(the object is built without __init__).  Everything else is the reference's code, executed as is."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402
from tests import inputfixture as fx  # noqa: E402


def main():
    ref_import.install_shims()
    cv2 = sys.modules["cv2"]
    cv2.INTER_LINEAR = 1

    def resize(img, size, interpolation=None):
        assert tuple(img.shape[:2]) == (size[1], size[0]), "only the identity resize occurs on this path"
        return img[:, :, 0] if img.ndim == 3 and img.shape[2] == 1 else img      # cv2 returns single-channel images 2-D
    cv2.resize = resize
    store = {}
    sys.modules["skvideo.io"].vread = lambda path: (_ for _ in ()).throw(IOError("no such video")) if store["frames"] is None else store["frames"]
    import importlib
    mod = importlib.import_module("datasets.ucf_dataloader")
    mod.vread = sys.modules["skvideo.io"].vread
    out = {}
    for k in range(fx.N_CASES):
        frames, ann, train = fx.case(k)
        store["frames"] = frames
        ds = object.__new__(mod.UCF101DataLoader)
        ds._dataset_dir = "DATA_PATH"; ds.name = "train" if train else "test"; ds._height = ds._width = 224
        ds.vid_files = [("v%d" % k, ann)]
        np.random.seed(1000 + k)
        s = ds[0]
        d = s["data"].numpy(); m = s["loc_msk"].numpy(); a = s["aug_data"].numpy()
        out["data_%d" % k] = d[:, :, ::9, ::7].astype(np.float64)
        out["aug_%d" % k] = a[:, :, ::9, ::7].astype(np.float64)
        out["mask_%d" % k] = np.packbits(m.astype(np.uint8))
        out["sums_%d" % k] = np.array([d.sum(), a.sum(), m.sum(), float(s["action"][0]), float(s["label_vid"])])
        out["dtypes_%d" % k] = np.array([str(s["data"].dtype), str(s["loc_msk"].dtype), str(s["aug_data"].dtype)])
    # JHMDB: __getitem__ after load_video (datasets/jhmdb_dataloader.py:102-230); load_video itself (cv2 decode / resize, .mat) is stubbed
    sys.modules["cv2"].INTER_AREA = 3; sys.modules["cv2"].INTER_NEAREST = 0
    jm = importlib.import_module("datasets.jhmdb_dataloader")
    for k in range(fx.N_JHMDB):
        frames, masks, label, ann, train = fx.jhmdb_case(k)
        ds = object.__new__(jm.JHMDB)
        ds.name = "train" if train else "test"; ds._height = ds._width = 224; ds.vid_files = ["v%d" % k]
        ds.load_video = lambda name, fr=frames, mk=masks, lb=label, an=ann: (fr.astype(np.float64), mk.copy(), lb, an)   # frames as the loader holds them: float64
        np.random.seed(2000 + k)
        s = ds[0]
        d = s["data"].numpy(); m = s["loc_msk"].numpy(); a = s["aug_data"].numpy(); mc = s["mask_cls"].numpy()
        out["jdata_%d" % k] = d[:, :, ::9, ::7].astype(np.float64)
        out["jaug_%d" % k] = a[:, :, ::9, ::7].astype(np.float64)
        out["jmask_%d" % k] = np.packbits(m.astype(np.uint8))
        out["jmcls_%d" % k] = mc[0, :, 0, 0].copy()
        out["jsums_%d" % k] = np.array([d.sum(), a.sum(), m.sum(), mc.sum(), float(s["action"][0])])
    path = os.path.join(ROOT, "tests", "golden", "input_pipe.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), [float(out["sums_%d" % k][2]) for k in range(fx.N_CASES)])


if __name__ == "__main__":
    main()
