"""The four networks of the reference as a plain PyTorch (ATen, CPU, fp32) op sequence, driven by the extracted weight
blob.  Test and baseline infrastructure only: it is what the reference's LibTorch CPU path executes
(feature/src/PPGExtractor.cpp:149-156 over net/Backbone.pt backbone.py:16-37, PointHeatmap.pt junction_decoder.py:9-15,
EdgeHeatmap.pt heatmap_decoder.py:8-22, Descriptor.pt descriptor_decoder.py:9-15) without the TorchScript archives,
which do not travel.  Never imported by the product path."""
import os
import struct

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_BLOB = os.path.join(os.path.dirname(HERE), "ppg-slam_amd", "weights", "ppg_weights.bin")


def read_blob(path=DEFAULT_BLOB):
    data = open(path, "rb").read()
    if data[:8] != b"PPGW0001":
        raise ValueError("not a PPG weight blob")
    n, _ = struct.unpack_from("<II", data, 8)
    out = {}
    for k in range(n):
        name, ndim, d0, d1, d2, d3, _pad, off, nbytes = struct.unpack_from("<56sI4IIQQ", data, 16 + 96 * k)
        dims = (d0, d1, d2, d3)[:ndim]
        out[name.rstrip(b"\0").decode()] = torch.from_numpy(np.frombuffer(data, np.float32, nbytes // 4, off).reshape(dims).copy())
    return out


class AtenNets:
    def __init__(self, blob=DEFAULT_BLOB):
        self.w = read_blob(blob)

    def _conv(self, x, name, relu=True, pad=1):
        y = F.conv2d(x, self.w[name + ".weight"], self.w[name + ".bias"], padding=pad)
        return F.relu(y) if relu else y

    @torch.no_grad()
    def inference(self, img_u8):
        """img_u8: [H, W] uint8 (numpy).  Returns feat, junc, heat, desc as NCHW tensors, like the reference."""
        x = torch.from_numpy(np.ascontiguousarray(img_u8)).to(torch.float32).div(255.0)[None, None]
        for blk, pool in (("1", True), ("2", True), ("3", True), ("4", False)):
            x = self._conv(x, f"backbone.conv{blk}a")
            x = self._conv(x, f"backbone.conv{blk}b")
            if pool:
                x = F.max_pool2d(x, 2, 2)
        feat = x
        junc = self._conv(self._conv(feat, "point.convPa"), "point.convPb", relu=False, pad=0)
        h = feat
        for k in range(3):
            p = f"edge.conv_block_lst.{k}"
            h = F.conv2d(h, self.w[p + ".0.weight"], self.w[p + ".0.bias"], padding=1)
            h = F.batch_norm(h, self.w[p + ".1.running_mean"], self.w[p + ".1.running_var"], self.w[p + ".1.weight"],
                             self.w[p + ".1.bias"], False, 0.0, 1e-5)
            h = F.pixel_shuffle(F.relu(h), 2)
        heat = F.conv2d(h, self.w["edge.conv_block_lst.3.weight"], self.w["edge.conv_block_lst.3.bias"])
        desc = self._conv(self._conv(feat, "desc.convDa"), "desc.convDb", relu=False, pad=0)
        return feat, junc, heat, desc
