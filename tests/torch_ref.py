"""PyTorch-CPU restatement of dlib's face_recognition_model_v1 forward (anet_type), used only to PIN the C oracle
(oracle/pvo_resnet.c) against an independent implementation.  [EXT] architecture: SURVEY.md appendix A.3."""
import numpy as np
import torch
import torch.nn.functional as F


def forward(chip_u8, params, units):
    """chip_u8: [150,150,3] uint8; params: dict name -> ndarray (models.split_resnet_blob); units: models.RESNET_UNITS"""
    t = lambda k: torch.from_numpy(np.ascontiguousarray(params[k])).double()
    avg = torch.tensor([122.782, 117.001, 104.298], dtype=torch.float32)
    x = ((torch.from_numpy(chip_u8.astype(np.float32)) - avg) / 256.0).permute(2, 0, 1)[None].double()
    x = F.conv2d(x, t("conv1.w"), t("conv1.b"), stride=2, padding=0)
    x = F.relu(x * t("aff1.g")[None, :, None, None] + t("aff1.b")[None, :, None, None])
    x = F.max_pool2d(x, 3, 2, 0)
    for u, (cin, n, down) in enumerate(units):
        a = F.conv2d(x, t("u%d.a.w" % u), t("u%d.a.b" % u), stride=2 if down else 1, padding=0 if down else 1)
        a = F.relu(a * t("u%d.a.g" % u)[None, :, None, None] + t("u%d.a.beta" % u)[None, :, None, None])
        b = F.conv2d(a, t("u%d.b.w" % u), t("u%d.b.b" % u), stride=1, padding=1)
        b = b * t("u%d.b.g" % u)[None, :, None, None] + t("u%d.b.beta" % u)[None, :, None, None]
        s = F.avg_pool2d(x, 2, 2, 0) if down else x
        oh, ow = max(b.shape[2], s.shape[2]), max(b.shape[3], s.shape[3])
        out = torch.zeros(1, n, oh, ow, dtype=torch.float64)
        out[:, :, :b.shape[2], :b.shape[3]] += b
        out[:, :s.shape[1], :s.shape[2], :s.shape[3]] += s
        x = F.relu(out)
    feat = x.mean(dim=(2, 3))[0]
    return (feat @ t("fc.w")).numpy()
