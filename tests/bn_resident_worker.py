"""Helper of test_gpu_ops.py::test_bn_backward_one_launch_self_service_gives_the_same_bits: runs the one-launch BatchNorm
backward on fixed seeded inputs and saves the results (the environment of the process decides whether workgroups wait for
their siblings or recompute the siblings' partial sums themselves)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)


def main(out):
    from ecg_hip import _lib as L
    L.load()
    res = {}
    for N, C, Lo, gap in ((256, 64, 500, False), (201, 128, 250, False), (256, 256, 125, True), (256, 128, 250, False),
                          (180, 100, 301, True)):
        g = torch.Generator().manual_seed(N + C + Lo)
        y = (torch.randn(N, C, Lo, generator=g) * 1.5 + 0.3).cuda()
        dp = torch.randn((N, C) if gap else (N, C, Lo // 2), generator=g).cuda()
        gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
        beta = (0.2 * torch.randn(C, generator=g)).cuda()
        mean = y.mean(dim=(0, 2))
        invstd = 1.0 / (y.var(dim=(0, 2), unbiased=False) + 1e-5).sqrt()
        ldy = (Lo + 63) // 64 * 64
        assert L.query("ecg_bn_relu_pool_bwd_launches", N, C, Lo, ldy) == 1
        ws = torch.empty(L.query("ecg_bn_relu_pool_bwd_ws_floats", N, C, Lo), device="cuda")
        for rep in range(2):                      # twice: the counters must come back clean
            dy = torch.full((N, C, ldy), float("nan"), device="cuda")
            dg, db = torch.full((C,), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda")
            L.call("ecg_bn_relu_pool_gap_bwd_ld" if gap else "ecg_bn_relu_pool_bwd_ld", L.f32(y), L.f32(dp), L.f32(gamma),
                   L.f32(beta), L.f32(mean), L.f32(invstd), L.f32(dy), ldy, L.f32(dg), L.f32(db), L.f32(ws), N, C, Lo, 1,
                   L.stream())
            torch.cuda.synchronize()
            key = f"{N}_{C}_{Lo}_{int(gap)}_{rep}"
            res["dy_" + key], res["dg_" + key], res["db_" + key] = dy.cpu().numpy(), dg.cpu().numpy(), db.cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
