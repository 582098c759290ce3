"""Helper of test_gpu_ops.py::test_bn_backward_one_launch_self_service_gives_the_same_bits: runs the one-launch BatchNorm
backward (ecg_bn_relu_pool_bwd_one_launch, caller-owned exchange words) on fixed seeded inputs and saves the results.
argv: <out.npz> [spin_polls]   (spin_polls = 0: no workgroup waits for its siblings — every one takes the self-service path)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)

CASES = ((256, 64, 500, False), (201, 128, 250, False), (256, 256, 125, True), (256, 128, 250, False), (180, 100, 301, True),
         (256, 32, 1000, False))          # 4, 2, 1, 2, 2 and 8 workgroups per channel


def run(spin_polls=-1):
    from ecg_hip import _lib as L
    L.load()
    res = {}
    for N, C, Lo, gap in CASES:
        g = torch.Generator().manual_seed(N + C + Lo)
        y = (torch.randn(N, C, Lo, generator=g) * 1.5 + 0.3).cuda()
        dp = torch.randn((N, C) if gap else (N, C, Lo // 2), generator=g).cuda()
        gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
        beta = (0.2 * torch.randn(C, generator=g)).cuda()
        mean = y.mean(dim=(0, 2))
        invstd = 1.0 / (y.var(dim=(0, 2), unbiased=False) + 1e-5).sqrt()
        ldy = (Lo + 63) // 64 * 64
        S = L.query("ecg_bn_relu_pool_bwd_one_launch_splits", N, C, Lo, ldy)
        assert S >= 1, (N, C, Lo)
        n_u = L.query("ecg_bn_relu_pool_bwd_one_launch_counter_uints", N, C, Lo, ldy)
        assert (n_u > 0) == (S > 1)
        cnt = torch.zeros(max(n_u, 2), dtype=torch.int32, device="cuda")      # OURS: zero once, reused by every launch
        for rep in range(2):                      # twice: the words must come back clean
            dy = torch.full((N, C, ldy), float("nan"), device="cuda")
            dg, db = torch.full((C,), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda")
            L.call("ecg_bn_relu_pool_bwd_one_launch", L.f32(y), L.f32(dp), L.f32(gamma), L.f32(beta), L.f32(mean),
                   L.f32(invstd), L.f32(dy), ldy, L.f32(dg), L.f32(db), L.ptr(cnt), N, C, Lo, 1, 1 if gap else 0,
                   spin_polls, L.stream())
            torch.cuda.synchronize()
            assert not bool(cnt.any()), "exchange words not returned to zero"
            key = f"{N}_{C}_{Lo}_{int(gap)}_{rep}"
            res["dy_" + key], res["dg_" + key], res["db_" + key] = dy.cpu().numpy(), dg.cpu().numpy(), db.cpu().numpy()
    return res


if __name__ == "__main__":
    np.savez(sys.argv[1], **run(int(sys.argv[2]) if len(sys.argv) > 2 else -1))
