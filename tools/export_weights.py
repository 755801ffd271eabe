#!/usr/bin/env python3
"""Extract the fp32 parameters of the four reference TorchScript nets into one flat blob.

Runs only in the build container (needs /root/reference/net/*.pt and torch).
Only the blob (data) is committed; the archives and this import never travel.

Blob layout (little endian):
    char[8]  magic  "PPGW0001"
    u32      n_tensors
    u32      reserved
    n_tensors x { char name[56]; u32 ndim; u32 dims[4]; u32 pad; u64 offset; u64 nbytes }   (96 B each)
    raw float32 tensors (OIHW for conv weights, as in state_dict), each 64-B aligned
"""
import struct
import sys
import numpy as np
import torch

NETS = {"backbone": "Backbone", "point": "PointHeatmap", "edge": "EdgeHeatmap", "desc": "Descriptor"}


def main(net_dir="/root/reference/net", out="ppg-slam_amd/weights/ppg_weights.bin"):
    tensors = []
    for short, fname in NETS.items():
        m = torch.jit.load(f"{net_dir}/{fname}.pt", map_location="cpu").eval()
        for k, v in m.state_dict().items():
            if v.dtype != torch.float32:
                continue  # num_batches_tracked
            tensors.append((f"{short}.{k}", v.detach().contiguous().numpy()))
    n = len(tensors)
    head = 16 + 96 * n
    off = (head + 63) // 64 * 64
    entries, blobs = [], []
    for name, a in tensors:
        dims = list(a.shape) + [0] * (4 - a.ndim)
        entries.append(struct.pack("<56sI4IIQQ", name.encode(), a.ndim, *dims, 0, off, a.nbytes))
        blobs.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    with open(out, "wb") as f:
        f.write(b"PPGW0001" + struct.pack("<II", n, 0))
        for e in entries:
            f.write(e)
        for o, b in blobs:
            f.seek(o)
            f.write(b)
        f.truncate(off)
    total = sum(a.size for _, a in tensors)
    print(f"wrote {out}: {n} tensors, {total} floats, {off} bytes")


if __name__ == "__main__":
    main(*sys.argv[1:])
