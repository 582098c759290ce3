# src/datasets/ptbxl_packed.py
"""GPU-side replacement for `DataLoader(PTBXLDataset(...))`, `DataLoader(PTBXLECGMultimodalDataset(...))`
and `DataLoader(PTBXLAFDataset(...))` (reference src/datasets/ptbxl.py, ptbxl_ecg_multimodal.py,
ptbxl_af.py; wired in scripts/03:95-117, 04:110-139, 05:92-116).

The reference's datasets stay importable next to this module (`src` is a namespace package); they
still build the split dataframe and the label matrix.  What changes is the per-item work: instead
of `wfdb.rdsamp` + numpy z-score per `__getitem__`, pack the split once and iterate device batches:

    ds = PTBXLECGMultimodalDataset(base_dir, "train", classes)            # reference: df + labels
    pack_split("train.ecgpack", base_dir, ds.df["filename_hr"], ds.y,
               demo=build_demo_matrix(r for _, r in ds.df.iterrows()), ids=ds.df.index.values)
    loader = make_loader("train.ecgpack", batch_size=64, shuffle=True)      # yields cuda tensors
    train_one_epoch_demo(model, loader, optimizer, device)                  # unchanged loop
"""
from ecg_hip.pack import (EcgPack, PackedBatchLoader, build_demo_matrix, build_demo_vector,  # noqa: F401
                          build_pack_from_wfdb, write_pack)


def pack_split(path, base_dir, rel_paths, labels, demo=None, ids=None):
    return build_pack_from_wfdb(path, base_dir, list(rel_paths), labels, demo, ids)


def make_loader(pack_path, batch_size, shuffle=False, seed=0, drop_last=False, device="cuda", rank=0,
                world_size=1, with_demo=None):
    return PackedBatchLoader(pack_path, batch_size, shuffle=shuffle, seed=seed, drop_last=drop_last,
                             device=device, rank=rank, world_size=world_size, with_demo=with_demo)
