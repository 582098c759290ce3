#!/usr/bin/env python3
"""Run a few train steps with a MARKER kernel in front of every ABI launch, and log the order of the entry
points — so that a rocprofv3 --pmc pass of this program can be split, dispatch by dispatch, into the entry points
bench.py prices (same keys: "<entry point>[int args]").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_step.py --log OUT/order.json

The marker is ecg_relu_fwd on 64 floats (kernel `relu_fwd_kernel`, never launched by the fused train path).
tools/make_profile_artifacts.py reads OUT/order.json + the counter CSVs of the passes.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["cnn", "multimodal"], default="cnn")
    ap.add_argument("--labels", type=int, default=5)
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--log", required=True)
    a = ap.parse_args()
    import torch
    from ecg_hip import _lib, functional as hipF
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    from src.utils.seed import set_seed
    hipF.set_conv_precision("bf16" if a.dtype == "bf16" else "fp32")
    dev = torch.device("cuda", 0)
    set_seed(42)
    demo = a.model == "multimodal"
    model = (ECGMultimodal(num_labels=a.labels) if demo else ECGCNN(num_labels=a.labels)).to(dev).train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(a.batch, 12, a.length, generator=g).to(dev)
    y = (torch.rand(a.batch, a.labels, generator=g) < 0.3).float().to(dev)
    xd = torch.rand(a.batch, 5, generator=g).to(dev)

    def step():
        opt.zero_grad()
        out = model(x, xd) if demo else model(x)
        hipF.backward_from_loss(hipF.binary_cross_entropy_with_logits(out, y))
        opt.step()

    step()                                   # unmarked warm-up (lazy loading, allocator)
    torch.cuda.synchronize()
    order = []
    m_in, m_out = torch.zeros(64, device=dev), torch.empty(64, device=dev)
    raw = _lib.call

    def marked(name, *args):
        raw("ecg_relu_fwd", m_in.data_ptr(), m_out.data_ptr(), 64, _lib.stream())
        order.append(f"{name}{[v for v in args[:-1] if isinstance(v, int) and abs(v) < (1 << 31)]}")
        raw(name, *args)

    _lib.call = hipF._call = marked
    import ecg_hip.optim as O
    for _ in range(a.steps):
        step()
    _lib.call = hipF._call = raw
    torch.cuda.synchronize()
    json.dump({"order": order, "args": vars(a)}, open(a.log, "w"))
    print(f"{len(order)} marked launches over {a.steps} steps")


if __name__ == "__main__":
    main()
