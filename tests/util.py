"""Shared helpers for the tests (input generators that follow the benchmark configurations)."""
import math

import torch


def stress_boxes(n: int, seed: int, side: float = 608.0, num_classes: int = 80):
    """NMS stress distribution: centres uniform in [0, side)^2, w,h log-uniform in [8, 400], XYXY fp32
    un-clipped; scores uniform in (0.001, 1); classes uniform."""
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 2, generator=g) * side
    wh = torch.exp(torch.rand(n, 2, generator=g) * (math.log(400.0) - math.log(8.0)) + math.log(8.0))
    boxes = torch.cat([c - wh / 2, c + wh / 2], dim=1).float()
    scores = (0.001 + torch.rand(n, generator=g) * 0.999).float()
    classes = torch.randint(0, num_classes, (n,), generator=g)
    return boxes, scores, classes


def clustered_boxes(n_clusters: int, per: int, seed: int, side: float = 608.0, num_classes: int = 80):
    """GT-like clusters of jittered boxes (deep suppression chains)."""
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n_clusters, 2, generator=g) * side
    wh = 20 + torch.rand(n_clusters, 2, generator=g) * 200
    cls = torch.randint(0, num_classes, (n_clusters,), generator=g)
    cc = c.repeat_interleave(per, 0) + torch.randn(n_clusters * per, 2, generator=g) * 6
    ww = wh.repeat_interleave(per, 0) * (1 + 0.1 * torch.randn(n_clusters * per, 2, generator=g))
    boxes = torch.cat([cc - ww / 2, cc + ww / 2], dim=1).float()
    scores = torch.rand(n_clusters * per, generator=g).float()
    return boxes, scores, cls.repeat_interleave(per, 0)


def random_heads(N: int, num_classes: int, grids, seed: int, obj_bias: float = -2.0, scale: float = 2.0,
                 num_anchors: int = 3):
    """Raw head tensors (NCHW f32) with a detector-like distribution: objectness logits shifted down so
    a small fraction of anchors fire."""
    g = torch.Generator().manual_seed(seed)
    K = 5 + num_classes
    heads = []
    for (gh, gw) in grids:
        t = torch.randn(N, num_anchors, K, gh, gw, generator=g) * scale
        t[:, :, 4] += obj_bias
        t[:, :, 2:4] *= 0.25  # keep exp() in range
        heads.append(t.reshape(N, num_anchors * K, gh, gw).contiguous())
    return heads
