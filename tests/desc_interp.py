"""Numpy interpreter of pc_conv_desc / pc_wgrad_desc semantics (include/picons.h), used to check
the host-side descriptor builders on CPU against torch.  Test infrastructure only."""
import numpy as np


def run_conv(d, x, w, bias=None, cscale=None, out=None):
    """x [N,Ti,Hi,Wi,ldi], w [Co, KT*KH*KW, ldw], out [N,To,Ho,Wo,ldo] (created if None)."""
    N = d["N"]
    if out is None:
        out = np.zeros((N, d["To"], d["Ho"], d["Wo"], d["ldo"]), np.float64)
    nt = d["ntap"]
    for tq in range(d["Tq"]):
        for hq in range(d["Hq"]):
            for wq in range(d["Wq"]):
                q = (tq, hq, wq)
                o = [q[i] * d["ostr"][i] + d["ooff"][i] for i in range(3)]
                acc = np.zeros((N, d["Co"]))
                for a in range(nt[0]):
                    for b in range(nt[1]):
                        for c in range(nt[2]):
                            abc = (a, b, c)
                            pos = [q[i] * d["istr"][i] + d["ioff0"][i] + abc[i] * d["istep"][i] for i in range(3)]
                            if not (0 <= pos[0] < d["Ti"] and 0 <= pos[1] < d["Hi"] and 0 <= pos[2] < d["Wi"]):
                                continue
                            wt = ((d["wk0"][0] + a * d["wkstep"][0]) * d["KH"] + d["wk0"][1] + b * d["wkstep"][1]) * d["KW"] \
                                + d["wk0"][2] + c * d["wkstep"][2]
                            xi = x[:, pos[0], pos[1], pos[2], :d["Ci"]]
                            acc += xi @ w[:, wt, :d["Ci"]].T
                if d["flags"] & 2:
                    acc = acc + bias
                if d["act"] == 1:
                    acc = np.maximum(acc, 0)
                elif d["act"] == 2:
                    acc = 1 / (1 + np.exp(-acc))
                if d["flags"] & 4:
                    acc = acc * cscale
                if d["flags"] & 1:
                    out[:, o[0], o[1], o[2], :d["Co"]] += acc
                else:
                    out[:, o[0], o[1], o[2], :d["Co"]] = acc
    return out


def run_wgrad(d, D, S):
    """D [N,Tq,Hq,Wq,ldd], S [N,Ts,Hs,Ws,lds] -> g [Cd, ntaps, Cs]."""
    nt = d["ntap"]
    g = np.zeros((d["Cd"], d["KT"] * d["KH"] * d["KW"], d["Cs"]))
    for tq in range(d["Tq"]):
        for hq in range(d["Hq"]):
            for wq in range(d["Wq"]):
                q = (tq, hq, wq)
                for a in range(nt[0]):
                    for b in range(nt[1]):
                        for c in range(nt[2]):
                            abc = (a, b, c)
                            tap = ((d["wk0"][0] + a) * d["KH"] + d["wk0"][1] + b) * d["KW"] + d["wk0"][2] + c
                            pos = [q[i] * d["istr"][i] + d["ioff0"][i] + abc[i] * d["istep"][i] for i in range(3)]
                            if 0 <= pos[0] < d["Ts"] and 0 <= pos[1] < d["Hs"] and 0 <= pos[2] < d["Ws"]:
                                g[:, tap, :] += D[:, tq, hq, wq, :d["Cd"]].T @ S[:, pos[0], pos[1], pos[2], :d["Cs"]]
    return g
