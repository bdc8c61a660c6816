"""Drop-in for /root/reference/utils/helpers.py: the bv / gv attentive masks, computed on the GPU
(pc_var_mask / pc_grad_mask) instead of the reference's host numpy round trip.  Same names, arguments
and output shapes; the result is a detached fp32 device tensor (callers cast it to
torch.cuda.FloatTensor right away, main_ucf101.py:117-118,131)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: E402,F401
from picons_amd import ops  # noqa: E402


def measure_pixelwise_var_v2(pred, flip_pred, frames_cnt=5, use_sig_output=False):
    if frames_cnt not in (3, 5):
        raise UnboundLocalError("frames_cnt must be 3 or 5 (utils/helpers.py:35-47 defines no other window)")
    return ops.var_mask(pred.detach().float().contiguous(), flip_pred.detach().float().contiguous(), frames_cnt, bool(use_sig_output))


def measure_pixelwise_gradient(pred, conf_thresh_lower=None, conf_thresh_upper=None):
    return ops.grad_mask(pred.detach().float().contiguous(), conf_thresh_lower, conf_thresh_upper)
