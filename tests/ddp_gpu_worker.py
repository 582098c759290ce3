"""Worker of tests/test_gpu_ddp.py: one rank of a 2-rank data-parallel step through the HIP path.
Launched by torch.distributed.run with ECG_HIP_REHEARSE_ON_ONE_GPU=1 (both ranks share device 0,
exchange over gloo).  Writes rank<r>.npz with the post-step state and the local loss."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]          # mode: "1"/"0" = FlatAdamW with/without overlap, "wrap" = FlatGradDDP
    overlap = mode == "1"
    from ecg_hip import ddp
    from ecg_hip.optim import FlatAdamW
    from oracle import ref_models as R
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    rank, world, local = ddp.init_distributed("nccl")
    assert world == 2 and local == 0
    dev = torch.device("cuda", 0)
    set_seed(42 + rank)                                   # different replicas: rank 0 must win
    model = ECGMultimodal().to(dev)
    ddp.broadcast_module_state(model, 0)
    step_model = model
    if mode == "wrap":                                    # stock optimizer + the gradient-averaging wrapper
        step_model = ddp.FlatGradDDP(model)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    else:
        opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, overlap=overlap)
    batch = R.synthetic_batch(16, 1000, 5, demo=True)     # global batch 16 -> 8 windows per rank
    shard = tuple(t.to(dev) for t in ddp.shard_batch(batch, rank, world))
    losses = [train_one_epoch_demo(step_model, [shard], opt, dev) for _ in range(2)]      # two steps
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.array(losses),
             **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
