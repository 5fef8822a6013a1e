"""GPU parity of the training-path kernels (batch-statistics BatchNorm + LeakyReLU forward/backward, conv weight
gradient, conv data gradient through yolo_conv2d_fwd, 2x dilation / 2x2 sum) against torch CPU autograd on the
same bf16-rounded operands -- the operations autograd runs for detectron2's Conv2d (F.conv2d + BatchNorm2d) and
F.leaky_relu in the reference's training step (yolo/modeling/backbone/darknet.py:113-126, fpn.py:33-37,66-71,96-100).

Tolerances: fp32 1e-3, bf16 5e-2 relative to the tensor's max (the task's conv/head bars), plus a mean-error bound.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(got, ref):
    return float((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 64, 13, 11), (2, 32, 40, 24), (2, 1024, 5, 3), (4, 256, 8, 8)],
                         ids=lambda s: "x".join(map(str, s)))
def test_bn_act_train_fwd_bwd(cuda, shape, dtype):
    from yolo_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    z = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).to(dtype).float()
    res = torch.randn(N, C, H, W, generator=g).to(dtype).float()
    gamma = 0.5 + torch.rand(C, generator=g)
    beta = 0.3 * torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    dy = torch.randn(N, C, H, W, generator=g).to(dtype).float()

    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.leaky_relu(F.batch_norm(zr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5), 0.1) + res
    y_ref.backward(dy)

    zv = ops.view_from_nchw(z.to(cuda), 1, dtype)
    rvw = ops.view_from_nchw(res.to(cuda), 0, dtype)
    yv = ops.alloc_view(N, H, W, C, 1, dtype, cuda)
    stats = torch.empty(5, C, device=cuda)
    ws = torch.empty(ops.bn_workspace_bytes(N, H, W, C), dtype=torch.uint8, device=cuda)
    rm_d, rv_d = rm.to(cuda), rv.to(cuda)
    ops.bn_train_stats(zv, gamma.to(cuda), beta.to(cuda), stats, ws, running_mean=rm_d, running_var=rv_d)
    ops.bn_act_fwd(zv, stats, yv, slope=0.1, residual=rvw)
    torch.cuda.synchronize()
    tol = 1e-3 if dtype == torch.float32 else 5e-2
    assert _rel(stats[0].cpu(), z.mean(dim=(0, 2, 3))) < 1e-4
    assert _rel(stats[1].cpu(), z.var(dim=(0, 2, 3), unbiased=False)) < 1e-4
    assert _rel(rm_d.cpu(), rm_ref) < 1e-4 and _rel(rv_d.cpu(), rv_ref) < 1e-4
    assert _rel(yv.nchw().float().cpu(), y_ref.detach()) < tol
    assert float(yv.buf[:, 0].abs().max()) == 0.0

    dyv = ops.view_from_nchw(dy.to(cuda), 0, dtype)
    dzv = ops.alloc_view(N, H, W, C, 1, dtype, cuda)
    dgam, dbet = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    ops.bn_act_bwd(dyv, zv, stats, dzv, dgam, dbet, ws, slope=0.1)
    torch.cuda.synchronize()
    assert _rel(dgam.cpu(), gr.grad) < 1e-3
    assert _rel(dbet.cpu(), br.grad) < 1e-3
    assert _rel(dzv.nchw().float().cpu(), zr.grad) < tol
    assert float(dzv.buf[:, :, 0].abs().max()) == 0.0


# (Cin, Cout, k, stride, H, W)
WGRAD_SHAPES = [(64, 128, 3, 1, 9, 7), (128, 64, 1, 1, 9, 7), (64, 128, 3, 2, 14, 10), (256, 512, 3, 1, 5, 4),
                (512, 255, 1, 1, 4, 4), (1024, 512, 1, 1, 3, 3), (32, 64, 3, 1, 13, 11), (384, 128, 1, 1, 6, 6),
                (128, 256, 3, 2, 10, 10), (256, 128, 1, 1, 26, 26)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("shape", WGRAD_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv_wgrad(cuda, shape, dtype):
    from yolo_amd import ops
    cin, cout, k, stride, H, W = shape
    g = torch.Generator().manual_seed(cin + 3 * cout + H)
    N = 3
    x = torch.randn(N, cin, H, W, generator=g).to(dtype).float()
    w = torch.randn(cout, cin, k, k, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2)
    dz = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(dz)
    Ho, Wo = y.shape[2], y.shape[3]
    xv = ops.view_from_nchw(x.to(cuda), 1, dtype)
    cpad = (cout + 7) // 8 * 8
    dzv = ops.alloc_view(N, Ho, Wo, cpad, 1, dtype, cuda)
    dzv.interior()[..., :cout].copy_(dz.to(cuda).permute(0, 2, 3, 1))
    if cpad != cout:
        dzv.interior()[..., cout:] = 7.0  # pad channels must not leak into the result
    dw = torch.full((cout, k, k, cin), 3.0, device=cuda)
    ws = torch.empty(ops.wgrad_workspace_bytes(N * Ho * Wo, cin, cout, k), dtype=torch.uint8, device=cuda)
    ops.conv2d_wgrad(xv, dzv, dw, ws, cin=cin, cout=cout, k=k, stride=stride)
    torch.cuda.synchronize()
    got = dw.permute(0, 3, 1, 2).cpu()
    assert _rel(got, w.grad) < 1e-3  # bf16 operands are exact inputs here; fp32 accumulation
    ops.conv2d_wgrad(xv, dzv, dw, ws, cin=cin, cout=cout, k=k, stride=stride, accumulate=True)
    torch.cuda.synchronize()
    assert _rel(dw.permute(0, 3, 1, 2).cpu(), 2 * w.grad) < 1e-3


@pytest.mark.parametrize("case", ["u8_bf16", "f32_f32", "u8_f32_padded"])
def test_stem_wgrad(cuda, case):
    """Weight gradient of the 3 -> 32 stem convolution straight from the images (normalised on the fly, zero beyond
    the stored image and at the borders) against torch autograd of F.conv2d on the normalised, padded image."""
    from yolo_amd import ops
    u8 = case.startswith("u8")
    dz_dtype = torch.bfloat16 if case.endswith("bf16") else torch.float32
    N, Hs, Ws = 3, 37, 45
    H, W = (64, 64) if case.endswith("padded") else (Hs, Ws)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (N, 3, Hs, Ws), generator=g, dtype=torch.uint8)
    img = img if u8 else img.float() + torch.rand(N, 3, Hs, Ws, generator=g)
    mean, std = [103.53, 116.28, 123.675], [57.375, 57.12, 58.395]
    xn = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    if dz_dtype == torch.bfloat16:
        xn = xn.bfloat16().float()
    xn = F.pad(xn, (0, W - Ws, 0, H - Hs))
    w = torch.zeros(32, 3, 3, 3, requires_grad=True)
    dz = torch.randn(N, 32, H, W, generator=g).to(dz_dtype).float()
    F.conv2d(xn, w, None, padding=1).backward(dz)
    dzv = ops.view_from_nchw(dz.to(cuda), 1, dz_dtype)
    dw = torch.full((32, 3, 3, 3), 5.0, device=cuda)
    ws = torch.empty(ops.stem_wgrad_workspace_bytes(), dtype=torch.uint8, device=cuda)
    kw = dict(mean=mean, std=std, round_input_bf16=dz_dtype == torch.bfloat16)
    ops.stem_wgrad(img.to(cuda), dzv, dw, ws, **kw)
    torch.cuda.synchronize()
    assert _rel(dw.permute(0, 3, 1, 2).cpu(), w.grad) < 1e-4
    ops.stem_wgrad(img.to(cuda), dzv, dw, ws, accumulate=True, **kw)
    torch.cuda.synchronize()
    assert _rel(dw.permute(0, 3, 1, 2).cpu(), 2 * w.grad) < 1e-4


@pytest.mark.parametrize("shape", [(64, 128, 3, 1, 9, 7), (128, 64, 1, 1, 9, 7), (64, 128, 3, 2, 14, 10),
                                   (512, 255, 1, 1, 4, 4), (128, 256, 3, 2, 10, 12)], ids=lambda s: "x".join(map(str, s)))
def test_conv_dgrad_via_fwd_kernel(cuda, shape):
    """dX = conv(dZ [dilated for stride 2], W transposed + rotated), with an accumulated second term."""
    from yolo_amd import ops
    cin, cout, k, stride, H, W = shape
    g = torch.Generator().manual_seed(cin + cout)
    N = 2
    x = torch.randn(N, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).bfloat16().float()
    y = F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2)
    dz = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dz)
    other = torch.randn(N, cin, H, W, generator=g).bfloat16().float()
    ref = x.grad + other
    Ho, Wo = y.shape[2], y.shape[3]
    cpad = (cout + 63) // 64 * 64  # the data-gradient conv reads Cout as ITS input channels
    dzv = ops.alloc_view(N, Ho, Wo, cpad, 1, torch.bfloat16, cuda)
    dzv.interior()[..., :cout].copy_(dz.to(cuda).permute(0, 2, 3, 1))
    wpad = torch.zeros(cpad, cin, k, k)
    wpad[:cout] = w
    wp, bp = ops.pack_conv_dgrad(wpad, torch.bfloat16, cuda)
    if stride == 2:
        big = ops.alloc_view(N, 2 * Ho, 2 * Wo, cpad, 1, torch.bfloat16, cuda)
        ops.dilate2x(dzv, big)
        src = big
        assert (2 * Ho, 2 * Wo) == (H, W)
    else:
        src = dzv
    dxv = ops.view_from_nchw(other.to(cuda), 1, torch.bfloat16)
    ops.conv2d_fwd(src, wp, bp, dxv, cin=cpad, cout=cin, k=k, stride=1, slope=1.0, residual=dxv)
    torch.cuda.synchronize()
    assert _rel(dxv.nchw().float().cpu(), ref) < 5e-2
    assert float((dxv.nchw().float().cpu() - ref).abs().mean() / ref.abs().mean()) < 4e-3


def test_dilate_and_sum2x2(cuda):
    from yolo_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 5, 7, generator=g).bfloat16().float()
    xv = ops.view_from_nchw(x.to(cuda), 1, torch.bfloat16)
    big = ops.alloc_view(2, 10, 14, 128, 1, torch.bfloat16, cuda)
    big.buf.fill_(5.0)
    ops.dilate2x(xv, big.slice(64, 64))
    torch.cuda.synchronize()
    got = big.nchw().float().cpu()
    ref = torch.zeros(2, 64, 10, 14)
    ref[:, :, ::2, ::2] = x
    assert torch.equal(got[:, 64:], ref) and bool((got[:, :64] == 5.0).all())
    up = torch.randn(2, 64, 10, 14, generator=g).bfloat16().float()
    uv = ops.view_from_nchw(up.to(cuda), 0, torch.bfloat16)
    sv = ops.alloc_view(2, 5, 7, 64, 1, torch.bfloat16, cuda)
    ops.sum2x2(uv, sv)
    torch.cuda.synchronize()
    ref = F.avg_pool2d(up, 2) * 4
    assert _rel(sv.nchw().float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pack_conv_train_batch(cuda, dtype):
    """One launch packs every layer: channels_last master weights -> the forward and data-gradient operand layouts
    that the torch-side packers (ops.pack_conv / ops.pack_conv_dgrad) produce."""
    from yolo_amd import ops
    torch.manual_seed(5)
    shapes = [(128, 64, 3), (64, 128, 1), (255, 256, 1), (256, 128, 3), (75, 512, 1)]
    entries, keep = [], []
    for cout, cin, k in shapes:
        w = torch.randn(cout, cin, k, k, device=cuda).contiguous(memory_format=torch.channels_last)
        if k == 1:  # channels_last is ambiguous for 1x1: build the storage explicitly
            w = torch.randn(cout, k, k, cin, device=cuda).permute(0, 3, 1, 2)
        cpad = (cout + 63) // 64 * 64
        fwd = torch.zeros(ops.lib().yolo_conv_cout_pad(cout), k * k * cin, dtype=dtype, device=cuda)
        dg = torch.zeros(ops.lib().yolo_conv_cout_pad(cin), k * k * cpad, dtype=dtype, device=cuda)
        entries.append((w, fwd, dg, cpad))
        keep.append((w, fwd, dg, cpad))
    tbl = ops.pack_table(entries, cuda)
    ops.pack_conv_train_batch(*tbl, dtype)
    torch.cuda.synchronize()
    for (cout, cin, k), (w, fwd, dg, cpad) in zip(shapes, keep):
        want_f, _ = ops.pack_conv(w.contiguous(), torch.zeros(cout), dtype, cuda)
        assert torch.equal(fwd[:cout].float().cpu(), want_f[:cout, :k * k * cin].float().cpu()), (cout, cin, k)
        assert float(fwd[cout:].abs().max()) == 0.0 if fwd.shape[0] > cout else True
        wpad = torch.zeros(cpad, cin, k, k, device=cuda)
        wpad[:cout] = w
        want_d, _ = ops.pack_conv_dgrad(wpad, dtype, cuda)
        assert torch.equal(dg[:cin].float().cpu(), want_d[:cin, :k * k * cpad].float().cpu()), (cout, cin, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 64, 128, 3, 1, 20, 12), (2, 128, 64, 1, 1, 13, 13), (2, 64, 256, 3, 2, 16, 16),
                                   (5, 128, 256, 1, 1, 9, 7), (87, 64, 128, 3, 1, 32, 32), (2, 64, 128, 3, 1, 52, 52),
                                   (4, 128, 256, 1, 1, 106, 106)], ids=lambda s: "x".join(map(str, s)))
def test_conv_with_bn_sums_equals_conv_then_stats(cuda, shape, dtype):
    """yolo_conv2d_fwd_bnsums + yolo_bn_train_finalize against yolo_conv2d_fwd + yolo_bn_train_stats: the same z
    bit for bit, the same statistics (sums of the stored z, folded in fp64 in both) and running-stat update."""
    from yolo_amd import ops
    N, cin, cout, k, s, H, W = shape
    torch.manual_seed(17)
    x = ops.alloc_view(N, H, W, cin, 1, dtype, cuda)
    x.interior().normal_()
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = ops.pack_conv(w, torch.zeros(cout), dtype, cuda)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    gamma, beta = torch.rand(cout, device=cuda) + 0.5, torch.randn(cout, device=cuda)
    res = []
    for fused in (False, True):
        z = ops.alloc_view(N, Ho, Wo, cout, 0, dtype, cuda)
        stats = torch.zeros(5, cout, device=cuda)
        rm, rv = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
        if fused:
            sums = torch.full((ops.conv_bnsums_rows(x, z, cin, cout, k, s) * 2 * cout,), float("nan"), device=cuda)
            rows = ops.conv2d_fwd_bnsums(x, wp, bp, z, sums, cin=cin, cout=cout, k=k, stride=s)
            ops.bn_train_finalize(sums, rows, N * Ho * Wo, gamma, beta, stats, eps=1e-5, momentum=0.1,
                                  running_mean=rm, running_var=rv)
        else:
            ops.conv2d_fwd(x, wp, bp, z, cin=cin, cout=cout, k=k, stride=s, slope=1.0)
            ws = torch.empty(ops.bn_workspace_bytes(N, Ho, Wo, cout), dtype=torch.uint8, device=cuda)
            ops.bn_train_stats(z, gamma, beta, stats, ws, eps=1e-5, momentum=0.1, running_mean=rm, running_var=rv)
        torch.cuda.synchronize()
        res.append((z.buf.clone(), stats, rm, rv))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.isfinite(b).all() and _rel(b, a) < 2e-5


# (N, Cin, Cout, k, H, W, form): form = what the launch is expected to pick, read off the number of rows of sums --
# "176": the 176-pixel tile, "strip": the 176-position strip form (rows of W + 2 positions), None: whatever fits
@pytest.mark.parametrize("shape", [(3, 128, 64, 3, 20, 12, None), (2, 64, 128, 1, 13, 13, None), (4, 256, 128, 3, 26, 26, None),
                                   (2, 128, 256, 1, 52, 52, None), (1, 64, 64, 3, 5, 3, None), (2, 256, 512, 3, 13, 13, None),
                                   (87, 64, 128, 3, 32, 32, "176"), (4, 128, 256, 1, 106, 106, "176"),
                                   (2, 64, 128, 3, 52, 52, "strip"), (5, 128, 256, 3, 41, 45, "strip")],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("with_residual", [False, True], ids=["plain", "accumulate"])
def test_dgrad_with_bn_backward_sums_equals_dgrad_then_bn_backward(cuda, shape, with_residual):
    """yolo_conv2d_fwd_bnbwd_sums + yolo_bn_act_bwd_presummed against yolo_conv2d_fwd + yolo_bn_act_bwd: the same dy bit
    for bit (the sums do not change what the convolution stores), the same dz to bf16 rounding and the same dgamma /
    dbeta (both reduce g and g * zhat over the stored dy and z; the orders of summation differ)."""
    from yolo_amd import ops
    N, cin, cout, k, H, W, form = shape   # the data-gradient convolution: cin -> cout; the BatchNorm has cout channels
    torch.manual_seed(31)
    dt = torch.bfloat16
    x = ops.alloc_view(N, H, W, cin, 1, dt, cuda)
    x.interior().normal_()
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = ops.pack_conv(w, torch.zeros(cout), dt, cuda)
    z = ops.alloc_view(N, H, W, cout, 0, dt, cuda)
    z.interior().normal_(0.3, 1.5)
    res = None
    if with_residual:
        res = ops.alloc_view(N, H, W, cout, 1, dt, cuda)
        res.interior().normal_()
    gamma, beta = torch.rand(cout, device=cuda) + 0.5, torch.randn(cout, device=cuda) * 0.5
    stats = torch.zeros(5, cout, device=cuda)
    ws = torch.empty(ops.bn_workspace_bytes(N, H, W, cout), dtype=torch.uint8, device=cuda)
    ops.bn_train_stats(z, gamma, beta, stats, ws, eps=1e-5, momentum=0.1)
    out = []
    for fused in (False, True):
        dy = ops.alloc_view(N, H, W, cout, 1, dt, cuda)
        dz = ops.alloc_view(N, H, W, cout, 1, dt, cuda)
        dgamma, dbeta = torch.zeros(cout, device=cuda), torch.zeros(cout, device=cuda)
        if fused:
            sums = torch.full((ops.conv_bnsums_rows(x, dy, cin, cout, k) * 2 * cout,), float("nan"), device=cuda)
            rows = ops.conv2d_fwd_bnbwd_sums(x, wp, bp, dy, sums, cin=cin, cout=cout, k=k, z=z, stats=stats,
                                             act_slope=0.1, residual=res)
            if form == "176":
                assert rows == -(-N * H * W // 176)
            elif form == "strip":
                assert rows == -(-N * H * (W + 2) // 176)
            ops.bn_act_bwd_presummed(dy, z, stats, dz, dgamma, dbeta, sums, rows, slope=0.1)
        else:
            ops.conv2d_fwd(x, wp, bp, dy, cin=cin, cout=cout, k=k, stride=1, slope=1.0, residual=res)
            ops.bn_act_bwd(dy, z, stats, dz, dgamma, dbeta, ws, slope=0.1)
        torch.cuda.synchronize()
        out.append((dy.buf.clone(), dz.buf.float().clone(), dgamma, dbeta))
    # the convolution may run another tile shape with the sums (the result of a tile does not depend on its shape)
    assert torch.equal(out[0][0], out[1][0])
    assert torch.isfinite(out[1][2]).all() and torch.isfinite(out[1][3]).all()
    assert _rel(out[1][3], out[0][3]) < 1e-4 and _rel(out[1][2], out[0][2]) < 1e-4
    assert float((out[1][1] - out[0][1]).abs().max()) <= 2 ** -7 * float(out[0][1].abs().max())
    assert float((out[1][1] != out[0][1]).float().mean()) < 0.02


@pytest.mark.parametrize("shape", [(2, 128, 256, 13, 13), (3, 256, 512, 7, 9), (1, 128, 128, 26, 20), (2, 512, 1024, 5, 5)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("with_residual", [False, True], ids=["plain", "accumulate"])
def test_stride2_dgrad_without_dilation(cuda, shape, with_residual):
    """yolo_conv3x3s2_dgrad against the form it replaces, yolo_dilate2x_nhwc + yolo_conv2d_fwd with the same
    data-gradient weights: the same products are summed in the same order (the dilated form only adds zeros), so
    the two agree bit for bit -- unless the stride-1 convolution of the dilated form runs in its strip form, which
    sums the taps in another order: then to one bf16 rounding; and against torch's conv_transpose2d on the
    bf16-rounded operands."""
    from yolo_amd import ops
    N, c_in, c_out, Hi, Wi = shape            # forward conv: c_in -> c_out, 3x3, stride 2; dz has c_out channels
    torch.manual_seed(23)
    w = torch.randn(c_out, c_in, 3, 3) * (2.0 / (c_in * 9)) ** 0.5
    cpad = (c_out + 63) // 64 * 64
    wpad = torch.zeros(cpad, c_in, 3, 3)
    wpad[:c_out] = w
    wd, bd = ops.pack_conv_dgrad(wpad.to(cuda), torch.bfloat16, cuda)
    dz = ops.alloc_view(N, Hi, Wi, cpad, 1, torch.bfloat16, cuda)
    dz.interior()[..., :c_out].normal_()
    other = ops.alloc_view(N, 2 * Hi, 2 * Wi, c_in, 1, torch.bfloat16, cuda)
    other.interior().normal_()
    res = other if with_residual else None
    dzd = ops.alloc_view(N, 2 * Hi, 2 * Wi, cpad, 1, torch.bfloat16, cuda)
    ops.dilate2x(dz, dzd)
    want = ops.alloc_view(N, 2 * Hi, 2 * Wi, c_in, 1, torch.bfloat16, cuda)
    ops.conv2d_fwd(dzd, wd, bd, want, cin=cpad, cout=c_in, k=3, stride=1, slope=1.0, residual=res)
    got = ops.alloc_view(N, 2 * Hi, 2 * Wi, c_in, 1, torch.bfloat16, cuda)
    got.buf.fill_(5.0)
    ops.conv3x3s2_dgrad(dz, wd, bd, got, cin=cpad, cout=c_in, residual=res)
    torch.cuda.synchronize()
    if "strip" in ops.conv_kernel_name(dzd, want, cpad, c_in, 3, 1):
        a, b = got.interior().float(), want.interior().float()
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
    else:
        assert torch.equal(got.interior(), want.interior())
    assert float(got.buf[:, 0].min()) == 5.0 and float(got.buf[:, :, 0].min()) == 5.0   # the halo is not written
    ref = F.conv_transpose2d(dz.nchw()[:, :c_out].float().cpu(), w.bfloat16().float(), stride=2, padding=1, output_padding=1)
    if with_residual:
        ref = ref + other.nchw().float().cpu()
    assert _rel(got.nchw().float().cpu(), ref) < 5e-2
