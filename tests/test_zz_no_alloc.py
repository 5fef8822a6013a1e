"""yolo_decode_noalloc (the entry point for callers that cannot allocate device memory: its workspace is a device
array built into the library) against yolo_decode, through the C ABI: two calls in a row, the second on a smaller
input, results bit for bit equal; sizes beyond the built-in scratch are refused with a clear error."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ANCHORS = (((10, 13), (16, 30), (33, 23)), ((30, 61), (62, 45), (59, 119)), ((116, 90), (156, 198), (373, 326)))


def _heads(N, S, C, seed, dev, dtype=torch.float32):
    from yolo_amd import ops
    g = torch.Generator().manual_seed(seed)
    ak = 3 * (5 + C)
    heads = []
    for s in (8, 16, 32):
        G = S // s
        v = ops.alloc_view(N, G, G, (ak + 7) // 8 * 8, 0, dtype, dev)
        t = torch.randn(N, G, G, ak, generator=g)
        t[..., 4::5 + C] -= 2.0  # objectness prior: a few per cent of the anchors pass
        v.buf[..., :ak].copy_(t.to(dev))
        heads.append(ops.View(v.buf, G, G, 0, ak, 0))
    return heads


def _cands(cap, N, dev):
    from yolo_amd import ops
    return ops.Candidates(torch.full((cap, 4), -7.0, device=dev), torch.full((cap,), -7.0, device=dev),
                          torch.full((cap,), -7, dtype=torch.int32, device=dev),
                          torch.full((N + 1,), -7, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_decode_noalloc_equals_decode_twice(cuda, dtype):
    from yolo_amd import ops
    total = 0
    for N, S, C, seed, thr in ((6, 256, 20, 1, 0.2), (2, 96, 20, 2, 0.1)):  # the second call is smaller
        heads = _heads(N, S, C, seed, cuda, dtype)
        cap = 1 << 16
        want = ops.decode(heads, (8, 16, 32), ANCHORS, C, thr, cap, out=_cands(cap, N, cuda))
        out = _cands(cap, N, cuda)
        held = torch.cuda.memory_allocated(cuda)
        got = ops.decode_noalloc(heads, (8, 16, 32), ANCHORS, C, thr, out)
        assert torch.cuda.memory_allocated(cuda) == held  # the call itself took nothing from torch's allocator
        torch.cuda.synchronize()
        m = int(want.img_offsets[N])
        assert 0 < m < cap
        total += m
        assert torch.equal(got.img_offsets, want.img_offsets)
        assert torch.equal(got.boxes[:m], want.boxes[:m])
        assert torch.equal(got.scores[:m], want.scores[:m])
        assert torch.equal(got.classes[:m], want.classes[:m])
        assert float(got.scores[m:].max()) == -7.0  # nothing written past the candidates
        del got, out, want  # (rebinding them in the next round would free memory during the measured call)
    assert total > 100


def test_decode_noalloc_refuses_what_does_not_fit(cuda):
    from yolo_amd import ops
    from yolo_amd._lib import YoloHipError
    N, S, C = 96, 608, 2   # 96 x 7581 cells = 727,776 slots > 524,288
    ak = 3 * (5 + C)
    heads = []
    for s in (8, 16, 32):
        G = S // s
        buf = torch.zeros((N, G, G, 24), device=cuda)
        heads.append(ops.View(buf, G, G, 0, ak, 0))
    with pytest.raises(YoloHipError, match="exceeds the built-in scratch"):
        ops.decode_noalloc(heads, (8, 16, 32), ANCHORS, C, 0.5, _cands(16, N, cuda))
