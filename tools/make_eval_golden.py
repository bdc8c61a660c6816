#!/usr/bin/env python3
"""AUTHORING CONTAINER ONLY: run the reference's own evaluation loop (/root/reference/evaluate_ucf101.py `iou`) on the
synthetic videos of tests/evalfixture.py with `FakeNet` standing in for the network, and record its accumulators in
tests/golden/eval_map.npz.  The data loader and the model are stubs; every line between the clip construction and the
f-mAP / v-mAP means (evaluate_ucf101.py:73-191) is the reference's, executed as is."""
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402
from tests import evalfixture  # noqa: E402


def main():
    ref_import.install_shims()
    np.int = int                                         # evaluate_ucf101.py:121 (removed from numpy 2)
    vids = evalfixture.videos()

    class DS(torch.utils.data.Dataset):
        def __init__(self, *a, **k):
            pass

        def __len__(self):
            return len(vids)

        def __getitem__(self, i):
            return vids[i]
    ref_import._stub("datasets.ucf_dataloader_eval", UCF101DataLoader=DS)
    ref_import._stub("models.capsules_ucf101", CapsNet=evalfixture.FakeNet)
    real_dl = torch.utils.data.DataLoader
    torch.utils.data.DataLoader = lambda dataset, **kw: real_dl(dataset, batch_size=kw.get("batch_size", 1), shuffle=False, num_workers=0)
    ck = tempfile.mkdtemp(prefix="picons_eval_", dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    open(os.path.join(ck, "best_model_train_0.pth"), "w").close()
    sys.argv = ["evaluate_ucf101.py", "--ckpt", ck]
    grabbed = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code.co_name == "iou":
            grabbed.update({k: frame.f_locals[k] for k in ("frame_ious", "video_ious", "n_tot_frames", "n_vids", "n_correct", "fmAP", "vmAP", "iou_threshs")})
    sys.setprofile(prof)
    try:
        runpy.run_path(os.path.join(ref_import.REF, "evaluate_ucf101.py"), run_name="reference_eval")
    finally:
        sys.setprofile(None)
        shutil.rmtree(ck, ignore_errors=True)
    out = {k: np.asarray(v) for k, v in grabbed.items()}
    out["threads"] = np.asarray(torch.get_num_threads())
    path = os.path.join(ROOT, "tests", "golden", "eval_map.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape, float(np.nansum(v))) for k, v in out.items()})


if __name__ == "__main__":
    main()
