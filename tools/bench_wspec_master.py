"""Times pc_wspec_master_planes (forward-layout and dgrad-layout launches) and pc_wspec_master_bwd at PrimaryCaps' shape (A = 544 = 512 + 32 rows,
B = 832, 9 x 9, 28-wide rows): the three HBM-bound kernels that turn the master OIHW weight into bf16 weight planes every step and the plane
gradients back.  Prints ms per launch pair and GB/s of the algorithmic bytes.  Also checks the planes / gradient against the first launch (bit-identical reruns).
    python tools/bench_wspec_master.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import capi, ops, spectral

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DEV = "cuda"
A1, A2, B, K, P = 512, 32, 832, 9, 28
A = A1 + A2
g = torch.Generator().manual_seed(3)
w1 = torch.randn(A1, B, K, K, generator=g).to(DEV); w2 = torch.randn(A2, B, K, K, generator=g).to(DEV)
m = {k: torch.from_numpy(v).to(DEV) for k, v in spectral.matrices(P, K).items()}
U, Ur, G = spectral.n_freq(P), len(spectral.freq_order(P)[1]), spectral.n_planes(P)
n = G * A * K * B
pf = torch.zeros(3 * n, dtype=torch.int16, device=DEV); pt = torch.zeros(3 * n, dtype=torch.int16, device=DEV)
dV = torch.randn(G, A, K, B, generator=g).to(DEV)
d1 = torch.empty(A1, B, K, K, device=DEV); d2 = torch.empty(A2, B, K, K, device=DEV)


def planes(of, ot):
    for w, cnt, a0 in ((w1, A1, 0), (w2, A2, A1)):
        capi.call("pc_wspec_master_planes", ops.ptr(w), ops.ptr(m["tw"]), cnt, a0, A, B, K, K, U, Ur, ops.ptr(of) if of is not None else None,
                  ops.ptr(ot) if ot is not None else None, n, ops.stream())


def bwd():
    ops.wspec_master_bwd(dV, m["tw"], A1, 0, A, B, K, K, U, Ur, d1)
    ops.wspec_master_bwd(dV, m["tw"], A2, A1, A, B, K, K, U, Ur, d2)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R * 1e3


planes(pf, pt); bwd(); torch.cuda.synchronize()
ref = (pf.clone(), pt.clone(), d1.clone(), d2.clone())
for name, fn, nbytes in (("planes, forward layout [g][A][KY][B]", lambda: planes(pf, None), A * B * 81 * 4 + 6 * n),
                         ("planes, dgrad layout [g][B][KY][A]", lambda: planes(None, pt), A * B * 81 * 4 + 6 * n),
                         ("plane gradients -> master gradient", bwd, 4 * n + A * B * 81 * 4)):
    ms = timeit(fn)
    print("%-42s %.3f ms  %.2f TB/s of %.0f MB" % (name, ms, nbytes / ms / 1e9, nbytes / 1e6), flush=True)
planes(pf, pt); bwd(); torch.cuda.synchronize()
print("bit-identical reruns:", all(torch.equal(a, b) for a, b in zip(ref, (pf, pt, d1, d2))),
      " checksums %d %d %.6e" % (int(pf.long().sum()), int(pt.long().sum()), float(d1.double().sum() + d2.double().sum())))
