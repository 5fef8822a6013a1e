"""GPU parity of the fused convolution (yolo_conv2d_fwd, yolo_stem_fwd, yolo_upsample2x_nhwc)
against plain torch fp32 CPU ops -- the routines the reference executes under detectron2's Conv2d
(yolo/modeling/backbone/darknet.py:113-126, fpn.py:33-37,66-71,96-100).

Tolerances (from the task statement): 1e-3 relative in fp32, 5e-2 relative in bf16.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3
BF16_TOL = 5e-2


def _rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    return float((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-12))


def _ref_conv(x, w, norm, bias, stride, slope, residual):
    k = w.shape[-1]
    y = F.conv2d(x, w, bias, stride=stride, padding=(k - 1) // 2)
    if norm is not None:
        y = F.batch_norm(y, norm["running_mean"], norm["running_var"], norm["weight"], norm["bias"], False, 0.1, 1e-5)
    if slope != 1.0:
        y = F.leaky_relu(y, slope)
    if residual is not None:
        y = y + residual
    return y


def _rand_norm(c, g):
    return dict(weight=0.5 + torch.rand(c, generator=g), bias=0.3 * torch.randn(c, generator=g),
                running_mean=0.3 * torch.randn(c, generator=g), running_var=0.5 + torch.rand(c, generator=g))


# (Cin, Cout, k, stride, H, W): the 23 distinct problem shapes of Darknet-53 + neck at small spatial
# crops (odd sizes on purpose: tiles with a ragged last pixel tile), incl. the unaligned 255/75 heads.
SHAPES = [
    (32, 64, 3, 2, 24, 20), (64, 32, 1, 1, 13, 11), (32, 64, 3, 1, 13, 11), (64, 128, 3, 2, 14, 10),
    (128, 64, 1, 1, 9, 7), (64, 128, 3, 1, 9, 7), (128, 256, 3, 2, 10, 10), (256, 128, 1, 1, 7, 5),
    (128, 256, 3, 1, 7, 5), (384, 128, 1, 1, 6, 6), (256, 255, 1, 1, 6, 5), (256, 512, 3, 2, 8, 6),
    (512, 256, 1, 1, 5, 5), (256, 512, 3, 1, 5, 4), (768, 256, 1, 1, 4, 4), (256, 128, 1, 1, 4, 3),
    (512, 255, 1, 1, 4, 4), (512, 1024, 3, 2, 6, 6), (1024, 512, 1, 1, 3, 3), (512, 1024, 3, 1, 3, 3),
    (512, 256, 1, 1, 3, 2), (1024, 255, 1, 1, 3, 3), (1024, 75, 1, 1, 3, 3),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv_shapes(cuda, shape, dtype):
    from yolo_amd import ops
    cin, cout, k, stride, H, W = shape
    g = torch.Generator().manual_seed(hash(shape) & 0xFFFF)
    N = 3
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    is_head = cout in (255, 75)
    norm = None if is_head else _rand_norm(cout, g)
    bias = 0.5 * torch.randn(cout, generator=g) if is_head else None
    slope = 1.0 if is_head else 0.1
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    use_res = (k == 3 and stride == 1 and not is_head)
    res = torch.randn(N, cout, Ho, Wo, generator=g) if use_res else None

    wf, bf = ops.fold_norm(w, bias, norm)
    if dtype == torch.bfloat16:  # the reference sees exactly what the kernel sees
        xq, rq = x.bfloat16().float(), (res.bfloat16().float() if res is not None else None)
        wq = wf.bfloat16().float()
        ref = _ref_conv(xq, wq, None, bf, stride, slope, rq)
    else:
        ref = _ref_conv(x, w, norm, bias, stride, slope, res)

    wp, bp = ops.pack_conv(wf, bf, dtype, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, dtype)
    cpad8 = (cout + 7) // 8 * 8
    yv = ops.alloc_view(N, Ho, Wo, cpad8, 1, dtype, cuda)
    rv = ops.view_from_nchw(res.to(cuda), 1, dtype) if res is not None else None
    ops.conv2d_fwd(xv, wp, bp, yv, cin=cin, cout=cout, k=k, stride=stride, slope=slope, residual=rv)
    torch.cuda.synchronize()
    got = yv.nchw()[:, :cout].float().cpu()
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    err = _rel_err(got, ref)
    assert err < tol, f"rel err {err}"
    if dtype == torch.bfloat16:
        # bf16 output rounding only: much tighter than the 5e-2 gate in the mean
        assert float((got - ref).abs().mean() / ref.abs().mean()) < 4e-3
    # halo untouched
    assert float(yv.buf[:, 0].abs().max()) == 0.0 and float(yv.buf[:, :, 0].abs().max()) == 0.0


# (Cin, Cout, N, H, W, residual, y halo, y channel offset / total): 3x3 stride-1 bf16 layers wide enough for the strip
# form (the three kw taps share one staged strip; pixels enumerated with their halo columns).  Tiles that span image
# rows and images, a ragged last tile, one image smaller than a tile, output into a channel slice, no halo in y.
STRIP_SHAPES = [
    (64, 128, 3, 9, 43, True, 1, (0, 128)), (128, 256, 2, 7, 41, True, 1, (128, 384)), (256, 512, 2, 30, 46, False, 1, (0, 512)),
    (128, 128, 5, 4, 40, True, 0, (0, 128)), (512, 1024, 1, 3, 60, False, 1, (0, 1024)), (64, 255, 2, 11, 53, False, 0, (0, 256)),
    (128, 256, 1, 52, 52, True, 1, (0, 256)),
]


@pytest.mark.parametrize("shape", STRIP_SHAPES, ids=lambda s: "x".join(map(str, s[:5])))
def test_conv_strip_form(cuda, shape):
    from yolo_amd import ops
    cin, cout, N, H, W, use_res, yhalo, (choff, chtot) = shape
    g = torch.Generator().manual_seed(cin + 3 * W)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    norm = _rand_norm(cout, g)
    res = torch.randn(N, cout, H, W, generator=g) if use_res else None
    wf, bf = ops.fold_norm(w, None, norm)
    ref = _ref_conv(x.bfloat16().float(), wf.bfloat16().float(), None, bf, 1, 0.1,
                    res.bfloat16().float() if use_res else None)
    wp, bp = ops.pack_conv(wf, bf, torch.bfloat16, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, torch.bfloat16)
    ybuf = ops.alloc_view(N, H, W, chtot, yhalo, torch.bfloat16, cuda)
    ybuf.buf.fill_(7.0)
    yv = ybuf.slice(choff, (cout + 7) // 8 * 8)
    rv = ops.view_from_nchw(res.to(cuda), 0, torch.bfloat16) if use_res else None
    assert "strip" in ops.conv_kernel_name(xv, yv, cin, cout, 3, 1)
    ops.conv2d_fwd(xv, wp, bp, yv, cin=cin, cout=cout, k=3, stride=1, residual=rv)
    torch.cuda.synchronize()
    got = yv.nchw()[:, :cout].float().cpu()
    assert _rel_err(got, ref) < BF16_TOL
    assert float((got - ref).abs().mean() / ref.abs().mean()) < 4e-3
    # nothing outside the output slice's interior was written: halo and the other channels keep the fill value
    full = ybuf.buf.float().cpu()
    if yhalo:
        assert bool((full[:, 0] == 7.0).all() and (full[:, -1] == 7.0).all())
        assert bool((full[:, :, 0] == 7.0).all() and (full[:, :, -1] == 7.0).all())
    keep = torch.ones(chtot, dtype=torch.bool)
    keep[choff:choff + (cout + 7) // 8 * 8] = False
    assert bool((full[..., keep] == 7.0).all())


# (Cin, Cout, k, stride, N, H_in, W_in, residual, y halo, y channel offset / total): launches that take the 128 x 176 tile
# (11 pixel fragment columns, the last one shared by the two pixel-waves) -- enough pixels that the tile count decides:
# ragged last tile, tiles spanning rows and images, residual (two uneven f32 staging halves) and plain (bf16 staging),
# stride 2, 1x1, two channel tiles, output into a channel slice without halo.
ODD_TILE_SHAPES = [
    (64, 128, 3, 1, 87, 32, 32, True, 1, (0, 128)), (64, 128, 3, 1, 87, 32, 32, False, 0, (128, 384)),
    (128, 256, 1, 1, 4, 106, 106, False, 1, (0, 256)), (64, 128, 3, 2, 87, 64, 64, False, 1, (0, 128)),
    (128, 256, 1, 1, 4, 106, 106, True, 1, (0, 256)),
]


@pytest.mark.parametrize("shape", ODD_TILE_SHAPES, ids=lambda s: "x".join(map(str, s[:8])))
def test_conv_176_pixel_tile(cuda, shape):
    from yolo_amd import ops
    cin, cout, k, stride, N, H, W, use_res, yhalo, (choff, chtot) = shape
    g = torch.Generator().manual_seed(cin + 5 * k + stride)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    norm = _rand_norm(cout, g)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(N, cout, Ho, Wo, generator=g) if use_res else None
    wf, bf = ops.fold_norm(w, None, norm)
    ref = _ref_conv(x.bfloat16().float(), wf.bfloat16().float(), None, bf, stride, 0.1,
                    res.bfloat16().float() if use_res else None)
    wp, bp = ops.pack_conv(wf, bf, torch.bfloat16, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, torch.bfloat16)
    ybuf = ops.alloc_view(N, Ho, Wo, chtot, yhalo, torch.bfloat16, cuda)
    ybuf.buf.fill_(7.0)
    yv = ybuf.slice(choff, cout)
    rv = ops.view_from_nchw(res.to(cuda), 0, torch.bfloat16) if use_res else None
    assert ",176," in ops.conv_kernel_name(xv, yv, cin, cout, k, stride), ops.conv_kernel_name(xv, yv, cin, cout, k, stride)
    ops.conv2d_fwd(xv, wp, bp, yv, cin=cin, cout=cout, k=k, stride=stride, residual=rv)
    torch.cuda.synchronize()
    got = yv.nchw()[:, :cout].float().cpu()
    assert _rel_err(got, ref) < BF16_TOL
    assert float((got - ref).abs().mean() / ref.abs().mean()) < 4e-3
    full = ybuf.buf.float().cpu()
    if yhalo:
        assert bool((full[:, 0] == 7.0).all() and (full[:, -1] == 7.0).all())
        assert bool((full[:, :, 0] == 7.0).all() and (full[:, :, -1] == 7.0).all())
    keep = torch.ones(chtot, dtype=torch.bool)
    keep[choff:choff + cout] = False
    assert bool((full[..., keep] == 7.0).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_concat_slice_and_upsample(cuda, dtype):
    """UpSampleBlock + torch.cat (fpn.py:33-37,179): conv output written 2x-upsampled into channels
    [0:C) of a wider buffer whose other channels must stay untouched; f32 output from bf16 input."""
    from yolo_amd import ops
    g = torch.Generator().manual_seed(5)
    N, cin, cout, H, W = 2, 256, 128, 5, 7
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    norm = _rand_norm(cout, g)
    wf, bf = ops.fold_norm(w, None, norm)
    xq = x.to(dtype).float()
    ref = F.interpolate(_ref_conv(xq, wf.to(dtype).float(), None, bf, 1, 0.1, None), scale_factor=2.0, mode="nearest")
    wp, bp = ops.pack_conv(wf, bf, dtype, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, dtype)
    cat = ops.alloc_view(N, 2 * H, 2 * W, 384, 1, dtype, cuda)
    cat.buf.fill_(7.0)
    ops.conv2d_fwd(xv, wp, bp, cat.slice(0, cout), cin=cin, cout=cout, k=1, stride=1, upsample2x=True)
    torch.cuda.synchronize()
    got = cat.nchw().float().cpu()
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    assert _rel_err(got[:, :cout], ref) < tol
    assert bool((got[:, cout:] == 7.0).all())
    # stand-alone upsample into a slice at a non-zero channel offset
    src = ops.view_from_nchw(x.to(cuda), 0, dtype)
    dst = ops.alloc_view(N, 2 * H, 2 * W, 512, 1, dtype, cuda)
    ops.upsample2x(src, dst.slice(128, cin))
    torch.cuda.synchronize()
    up = F.interpolate(xq, scale_factor=2.0, mode="nearest")
    assert torch.equal(dst.nchw()[:, 128:128 + cin].float().cpu(), up)
    assert float(dst.nchw()[:, :128].abs().max()) == 0.0


def test_conv_f32_out_from_bf16(cuda):
    from yolo_amd import ops
    g = torch.Generator().manual_seed(9)
    N, cin, cout, H, W = 2, 512, 255, 4, 4
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b)
    wp, bp = ops.pack_conv(w, b, torch.bfloat16, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, torch.bfloat16)
    yv = ops.alloc_view(N, H, W, 256, 0, torch.float32, cuda)
    ops.conv2d_fwd(xv, wp, bp, yv, cin=cin, cout=cout, k=1, stride=1, slope=1.0)
    torch.cuda.synchronize()
    assert _rel_err(yv.nchw()[:, :cout].cpu(), ref) < 1e-4  # f32 accumulate, f32 store


@pytest.mark.parametrize("img_dtype", [torch.float32, torch.uint8], ids=["f32img", "u8img"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stem(cuda, dtype, img_dtype):
    """preprocess ((x - mean)/std, zero pad to a multiple of 32) + BasicStem (darknet.py:57-60)."""
    from yolo_amd import ops
    g = torch.Generator().manual_seed(3)
    N, Hs, Ws, H, W = 2, 50, 75, 64, 96
    img = torch.randint(0, 256, (N, 3, Hs, Ws), generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * (2.0 / (32 * 9)) ** 0.5
    norm = _rand_norm(32, g)
    mean, std = (10.0, 20.0, 30.0), (255.0, 128.0, 64.0)
    xin = torch.zeros(N, 3, H, W)
    xin[:, :, :Hs, :Ws] = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    ref = _ref_conv(xin, w, norm, None, 1, 0.1, None)
    wf, bf = ops.fold_norm(w, None, norm)
    ws, bs = ops.pack_stem(wf, bf, cuda)
    yv = ops.alloc_view(N, H, W, 32, 1, dtype, cuda)
    ops.stem_fwd(img.to(img_dtype).to(cuda).contiguous(), ws, bs, yv, mean=mean, std=std)
    torch.cuda.synchronize()
    got = yv.nchw().float().cpu()
    assert _rel_err(got, ref) < (FP32_TOL if dtype == torch.float32 else BF16_TOL)


def test_conv_rejects_bad_arguments(cuda):
    from yolo_amd import ops
    from yolo_amd._lib import YoloHipError
    x = ops.alloc_view(1, 4, 4, 64, 0, torch.bfloat16, cuda)  # halo 0 < pad 1
    y = ops.alloc_view(1, 4, 4, 64, 1, torch.bfloat16, cuda)
    w = torch.zeros(128, 9 * 64, dtype=torch.bfloat16, device=cuda)
    b = torch.zeros(128, device=cuda)
    with pytest.raises(YoloHipError, match="halo"):
        ops.conv2d_fwd(x, w, b, y, cin=64, cout=64, k=3, stride=1)
    x2 = ops.alloc_view(1, 4, 4, 48, 1, torch.bfloat16, cuda)
    with pytest.raises(YoloHipError, match="Cin"):
        ops.conv2d_fwd(x2, w, b, y, cin=48, cout=64, k=1, stride=1)


@pytest.mark.parametrize("img_dtype", [torch.float32, torch.uint8], ids=["f32img", "u8img"])
@pytest.mark.parametrize("geom", [(2, 50, 75, 64, 96), (3, 96, 160, 96, 160), (1, 33, 47, 64, 64)],
                         ids=lambda g_: "x".join(map(str, g_)))
def test_stem_down_fused(cuda, geom, img_dtype):
    """yolo_stem_down_fwd = preprocess + BasicStem (darknet.py:57-60) + res1.downsample (darknet.py:113-116),
    against torch fp32 ops on bf16-rounded weights and a bf16-rounded stem output (what the two separate
    bf16 launches compute), and bit-for-bit-close against those two launches themselves."""
    from yolo_amd import ops
    g = torch.Generator().manual_seed(13)
    N, Hs, Ws, H, W = geom
    img = torch.randint(0, 256, (N, 3, Hs, Ws), generator=g)
    w0 = torch.randn(32, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    n0 = _rand_norm(32, g)
    w1 = torch.randn(64, 32, 3, 3, generator=g) * (2.0 / (32 * 9)) ** 0.5
    n1 = _rand_norm(64, g)
    mean, std = (10.0, 20.0, 30.0), (255.0, 128.0, 64.0)
    xin = torch.zeros(N, 3, H, W)
    xin[:, :, :Hs, :Ws] = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    wf0, bf0 = ops.fold_norm(w0, None, n0)
    wf1, bf1 = ops.fold_norm(w1, None, n1)
    s = _ref_conv(xin.bfloat16().float(), wf0.bfloat16().float(), None, bf0, 1, 0.1, None).bfloat16().float()
    ref = _ref_conv(s, wf1.bfloat16().float(), None, bf1, 2, 0.1, None)

    ws, bs = ops.pack_stem(wf0, bf0, cuda)
    wd, bd = ops.pack_conv(wf1, bf1, torch.bfloat16, cuda)
    dimg = img.to(img_dtype).to(cuda).contiguous()
    yv = ops.alloc_view(N, H // 2, W // 2, 64, 1, torch.bfloat16, cuda)
    ops.stem_down_fwd(dimg, ws, bs, wd, bd, yv, padded_hw=(H, W), mean=mean, std=std)
    torch.cuda.synchronize()
    got = yv.nchw().float().cpu()
    assert _rel_err(got, ref) < BF16_TOL
    assert float((got - ref).abs().mean() / ref.abs().mean()) < 4e-3
    assert float(yv.buf[:, 0].abs().max()) == 0.0 and float(yv.buf[:, :, -1].abs().max()) == 0.0

    sv = ops.alloc_view(N, H, W, 32, 1, torch.bfloat16, cuda)
    y2 = ops.alloc_view(N, H // 2, W // 2, 64, 1, torch.bfloat16, cuda)
    ops.stem_fwd(dimg, ws, bs, sv, mean=mean, std=std)
    ops.conv2d_fwd(sv, wd, bd, y2, cin=32, cout=64, k=3, stride=2)
    torch.cuda.synchronize()
    two = y2.nchw().float().cpu()
    # same bf16 stem values, fp32 accumulation in a different order: at most one bf16 ulp apart
    assert float((got - two).abs().max() / two.abs().max()) < 2 ** -7
    assert float((got != two).float().mean()) < 0.05


@pytest.mark.parametrize("geom", [(2, 24, 40), (3, 13, 21), (1, 8, 16), (2, 52, 52)], ids=lambda g_: "x".join(map(str, g_)))
def test_resblock_c64_fused(cuda, geom):
    """yolo_resblock_c64_fwd = ResidualBlock.forward minus the downsample (darknet.py:117-126) for C = 64, against
    torch fp32 ops on bf16-rounded operands with a bf16-rounded intermediate, and against the two separate
    launches; ragged tiles, image borders (zero padding of the intermediate) and the halo stay untouched."""
    from yolo_amd import ops
    g = torch.Generator().manual_seed(29)
    N, H, W = geom
    x = torch.randn(N, 64, H, W, generator=g)
    w1 = torch.randn(32, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    w2 = torch.randn(64, 32, 3, 3, generator=g) * (2.0 / (32 * 9)) ** 0.5
    n1, n2 = _rand_norm(32, g), _rand_norm(64, g)
    wf1, bf1 = ops.fold_norm(w1, None, n1)
    wf2, bf2 = ops.fold_norm(w2, None, n2)
    xq = x.bfloat16().float()
    t = _ref_conv(xq, wf1.bfloat16().float(), None, bf1, 1, 0.1, None).bfloat16().float()
    ref = _ref_conv(t, wf2.bfloat16().float(), None, bf2, 1, 0.1, xq)

    wp1, bp1 = ops.pack_conv(wf1, bf1, torch.bfloat16, cuda)
    wp2, bp2 = ops.pack_conv(wf2, bf2, torch.bfloat16, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, torch.bfloat16)
    yv = ops.alloc_view(N, H, W, 64, 1, torch.bfloat16, cuda)
    ops.resblock_c64_fwd(xv, wp1, bp1, wp2, bp2, yv)
    torch.cuda.synchronize()
    got = yv.nchw().float().cpu()
    assert _rel_err(got, ref) < BF16_TOL
    assert float((got - ref).abs().mean() / ref.abs().mean()) < 4e-3
    assert float(yv.buf[:, 0].abs().max()) == 0.0 and float(yv.buf[:, :, -1].abs().max()) == 0.0
    assert float(yv.buf[:, -1].abs().max()) == 0.0 and float(yv.buf[:, :, 0].abs().max()) == 0.0

    tv = ops.alloc_view(N, H, W, 32, 1, torch.bfloat16, cuda)
    y2 = ops.alloc_view(N, H, W, 64, 1, torch.bfloat16, cuda)
    ops.conv2d_fwd(xv, wp1, bp1, tv, cin=64, cout=32, k=1, stride=1)
    ops.conv2d_fwd(tv, wp2, bp2, y2, cin=32, cout=64, k=3, stride=1, residual=xv)
    torch.cuda.synchronize()
    two = y2.nchw().float().cpu()
    assert float((got - two).abs().max() / two.abs().max()) < 2 ** -7
    assert float((got != two).float().mean()) < 0.05
    # in-place use is rejected
    from yolo_amd._lib import YoloHipError
    with pytest.raises(YoloHipError, match="overlap"):
        ops.resblock_c64_fwd(xv, wp1, bp1, wp2, bp2, xv)


# (Cin, Cout, k, stride, N, H, W, residual, upsample): few-pixel layers, where the k loop of a tile is split over
# several workgroups (single-image inference)
SPLITK_SHAPES = [(512, 1024, 3, 1, 1, 13, 13, True, False), (256, 512, 3, 1, 1, 26, 26, True, False),
                 (1024, 512, 1, 1, 1, 13, 13, False, False), (128, 256, 3, 1, 1, 52, 52, False, False),
                 (768, 256, 1, 1, 2, 13, 13, False, True), (1024, 255, 1, 1, 1, 13, 13, False, False),
                 (256, 512, 3, 2, 1, 52, 52, False, False), (768, 256, 1, 1, 1, 26, 26, False, False)]


@pytest.mark.parametrize("shape", SPLITK_SHAPES, ids=lambda s: "x".join(map(str, s[:7])))
def test_conv_splitk(cuda, shape):
    """yolo_conv2d_fwd_splitk against the torch reference and against yolo_conv2d_fwd (equal to bf16 rounding)."""
    from yolo_amd import ops
    cin, cout, k, stride, N, H, W, use_res, up = shape
    g = torch.Generator().manual_seed(cin + W)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    is_head = cout == 255
    norm = None if is_head else _rand_norm(cout, g)
    bias = 0.5 * torch.randn(cout, generator=g) if is_head else None
    slope = 1.0 if is_head else 0.1
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(N, cout, Ho, Wo, generator=g) if use_res else None
    wf, bf = ops.fold_norm(w, bias, norm)
    ref = _ref_conv(x.bfloat16().float(), wf.bfloat16().float(), None, bf, stride, slope,
                    res.bfloat16().float() if use_res else None)
    if up:
        ref = F.interpolate(ref, scale_factor=2.0, mode="nearest")
    wp, bp = ops.pack_conv(wf, bf, torch.bfloat16, cuda)
    xv = ops.view_from_nchw(x.to(cuda), 1, torch.bfloat16)
    ydt = torch.float32 if is_head else torch.bfloat16
    f = 2 if up else 1
    outs = []
    splits, nbytes = ops.conv_splitk_plan(N * Ho * Wo, cin, cout, k, stride, torch.bfloat16)
    assert splits > 1 and nbytes > 0
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=cuda)  # NaN bit patterns: nothing unwritten is read
    for split in (True, False):
        yv = ops.alloc_view(N, Ho * f, Wo * f, (cout + 7) // 8 * 8, 1, ydt, cuda)
        rv = ops.view_from_nchw(res.to(cuda), 1, torch.bfloat16) if use_res else None
        if split:
            ops.conv2d_fwd_splitk(xv, wp, bp, yv, ws, cin=cin, cout=cout, k=k, stride=stride, slope=slope, residual=rv,
                                  upsample2x=up)
        else:
            ops.conv2d_fwd(xv, wp, bp, yv, cin=cin, cout=cout, k=k, stride=stride, slope=slope, residual=rv, upsample2x=up)
        torch.cuda.synchronize()
        outs.append(yv.nchw()[:, :cout].float().cpu())
        assert float(yv.buf[:, 0].abs().max()) == 0.0 and float(yv.buf[:, :, 0].abs().max()) == 0.0
    assert _rel_err(outs[0], ref) < (1e-3 if is_head else BF16_TOL)
    assert float((outs[0] - ref).abs().mean() / ref.abs().mean()) < 4e-3
    assert _rel_err(outs[0], outs[1]) < 2.0 ** -7
