"""GPU parity of one whole training step (forward in training mode, the three losses, backward through the HIP
plan) against the oracle's autograd restatement of the reference (oracle/yolo_train_ref.py: DenseDetector.forward
-> YOLOV3.forward_training, yolo/modeling/meta_arch/yolo.py:101-264) on the same state dict, images and targets.

Why two kinds of test: LeakyReLU's derivative jumps at 0, so a 1e-5 difference in a pre-activation flips single
gradient elements between 1 and 0.1.  In fp32 the HIP plan and the oracle agree to ~1e-5 in every activation and
the gradients to ~1 % (measured: relative L2 0.4-2 %, cosine > 0.9998).  In bf16 on this randomly initialised
network the forward already differs by 1-25 % at the heads (rounding doubles per stage), which decorrelates the
gradients of an fp32 oracle from ANY bf16 run; the bf16 plan is therefore checked layer by layer against torch
autograd applied to the plan's own tensors (every z, y, dz, dW and every accumulated activation gradient).
"""
import pytest
import torch
import torch.nn.functional as F

import workload

pytestmark = pytest.mark.gpu


def _make(dtype, N=4, S=128, freeze_at=2):
    from oracle import yolo_train_ref as T
    from yolo_amd.structures import Boxes, Instances
    model, cfg = workload.build_model(config="yolov3.yaml", dtype=dtype, opts=["MODEL.BACKBONE.FREEZE_AT", freeze_at])
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    imgs = workload.synthetic_images(N, S, S, seed=3)
    gb, gc = T.synthetic_targets(N, S, C, seed=11, max_boxes=6)
    inputs = [dict(image=im, instances=Instances((S, S), gt_boxes=Boxes(b), gt_classes=c)) for im, b, c in zip(imgs, gb, gc)]
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return model, cfg, sd, imgs, gb, gc, inputs


def _l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("freeze_at", [2, 1, 0])
def test_training_step_fp32_matches_oracle(cuda, freeze_at):
    """MODEL.BACKBONE.FREEZE_AT = 2 is the reference's default; 1 trains the first stage too, 0 the stem as well
    (DarkNet.freeze, darknet.py:236-258)."""
    from oracle import yolo_train_ref as T
    model, cfg, sd, imgs, gb, gc, inputs = _make("f32", freeze_at=freeze_at)
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    full = dict(sd)
    full.update(params)
    stats, heads_ref = {}, {}
    want = T.forward_training(imgs.float() / 255.0, full, gb, gc, C, freeze_at=freeze_at, stats=stats, heads_out=heads_ref)
    sum(want.values()).backward()
    model.train()
    got = model(inputs)
    sum(got.values()).backward()
    torch.cuda.synchronize()
    for k in want:
        assert abs(float(got[k].detach()) - float(want[k].detach())) / abs(float(want[k].detach())) < 1e-3, k
    P = model.backbone.train_engine().plan(len(inputs), imgs.shape[2], imgs.shape[3])
    for name, hv in zip(["p3", "p4", "p5"], P.heads):
        ref = heads_ref[name].detach()
        assert float((hv.nchw().float().cpu() - ref).abs().max() / ref.abs().max()) < 1e-3, name
    frozen = set(T.frozen_names(freeze_at))
    n = 0
    for name, p in model.named_parameters():
        layer = name.rsplit(".norm.", 1)[0].rsplit(".weight", 1)[0].rsplit(".bias", 1)[0]
        if layer in frozen:
            assert p.grad is None and not p.requires_grad
            continue
        g, ref = p.grad.float().cpu(), params[name].grad
        cos = float(F.cosine_similarity(g.flatten().double(), ref.flatten().double(), dim=0))
        assert cos > 0.998 and _l2(g, ref) < 0.06, (name, cos, _l2(g, ref))
        n += 1
    assert n == {2: 210, 1: 219, 0: 222}[freeze_at]   # + 3 convolutions of res1 (weight, gamma, beta each), + the stem
    # running statistics moved exactly as torch's BatchNorm2d moves them
    bufs = dict(model.named_buffers())
    moved = ["backbone.fpn_lateral0.blocks.0", "backbone.bottom_up.res3.2.conv2"]
    moved += ["backbone.bottom_up.res1.0.conv1"] if freeze_at <= 1 else []
    moved += ["backbone.bottom_up.stem.conv"] if freeze_at == 0 else []
    for name in moved:
        assert _l2(bufs[name + ".norm.running_mean"].cpu(), stats[name][0]) < 1e-3
        assert _l2(bufs[name + ".norm.running_var"].cpu(), stats[name][1]) < 1e-3
    for name in ["backbone.bottom_up.stem.conv"] + (["backbone.bottom_up.res1.0.conv1"] if freeze_at >= 2 else []):
        if freeze_at >= 1:
            assert torch.equal(bufs[name + ".norm.running_mean"].cpu(), sd[name + ".norm.running_mean"])
    model.eval()


def _consumers(P):
    """activation buffer id -> list of (layer, kind) that read it: 'conv' input or 'skip' (residual add)."""
    out = {}
    for L in P.layers:
        if L.stem:
            continue  # reads the images
        out.setdefault(id(L.x.buf), []).append((L, "conv"))
        if L.residual is not None:
            out.setdefault(id(L.residual.buf), []).append((L, "skip"))
    return out


@pytest.mark.parametrize("dtype,freeze_at", [("bf16", 2), ("f32", 2), ("bf16", 0), ("bf16", 1)])
def test_training_plan_is_self_consistent(cuda, dtype, freeze_at):
    """Every step of the plan against torch CPU autograd on the plan's OWN tensors: forward conv / BatchNorm / leaky /
    residual of each trainable layer, its dz, dW, dgamma, dbeta, and the accumulated gradient of every activation
    (sum over all consumers, skip connections, the 2x upsample and the channel concat).  FREEZE_AT 1 and 0 add the
    thin first-stage layers and the stem (which reads the images and has its own forward and weight-gradient kernels)."""
    model, cfg, sd, imgs, gb, gc, inputs = _make(dtype, N=2, S=128, freeze_at=freeze_at)
    model.train()
    got = model(inputs)
    sum(got.values()).backward()
    torch.cuda.synchronize()
    eng = model.backbone.train_engine()
    P = eng.plan(2, 128, 128)
    tol = 5e-2 if dtype == "bf16" else 1e-3
    q = (lambda t: t.bfloat16().float()) if dtype == "bf16" else (lambda t: t)
    cons = _consumers(P)
    checked_dx = 0
    for L in P.layers:
        if L.frozen:
            continue
        c = L.conv
        if L.stem:   # the normalised image, as the stem kernel forms it
            mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(1, 3, 1, 1)
            std = torch.tensor(cfg.MODEL.PIXEL_STD).view(1, 3, 1, 1)
            x = q((imgs.float() - mean) / std)
        else:
            x = L.x.nchw().float().cpu()
        w = q(c.weight.detach().float().cpu())
        pad = (c.kernel_size - 1) // 2
        if c.norm is None:   # head: conv + bias, f32 output
            ref = F.conv2d(x, w, c.bias.detach().cpu(), stride=1, padding=pad)
            assert float((L.y.nchw()[:, :c.out_channels].cpu() - ref).abs().max() / ref.abs().max()) < tol, L.name
            dz = L.dz.nchw()[:, :c.out_channels].float().cpu()
        else:
            z = L.z.nchw().float().cpu()
            zref = F.conv2d(x, w, None, stride=c.stride, padding=pad)
            assert float((z - zref).abs().max() / zref.abs().max()) < tol, (L.name, "z")
            zz = z.clone().requires_grad_(True)
            gam = c.norm.weight.detach().cpu().clone().requires_grad_(True)
            bet = c.norm.bias.detach().cpu().clone().requires_grad_(True)
            yy = F.leaky_relu(F.batch_norm(zz, None, None, gam, bet, True, 0.1, 1e-5), 0.1)
            yref = yy.detach() + (L.residual.nchw().float().cpu() if L.residual is not None else 0)
            y = L.y.nchw().float().cpu()
            assert float((y - yref).abs().max() / yref.abs().max()) < tol, (L.name, "y")
            if L.up_dst is not None:
                up = L.up_dst.nchw().float().cpu()
                assert torch.equal(up, F.interpolate(y, scale_factor=2.0, mode="nearest")), (L.name, "upsample")
            dy = L.dy.nchw().float().cpu()
            yy.backward(dy)
            dz = L.dz.nchw().float().cpu()
            # relative L2: a pre-activation within rounding of 0 may take the other LeakyReLU slope in one element
            assert _l2(dz, zz.grad) < tol / 2, (L.name, "dz", _l2(dz, zz.grad))
            ptol = 2.5e-2 if dtype == "bf16" else 2e-3   # same caveat: elements with z == batch mean sit on the kink
            assert _l2(L.dgamma.cpu(), gam.grad) < ptol and _l2(L.dbeta.cpu(), bet.grad) < ptol, (L.name, "dgamma/dbeta")
        # weight gradient from the plan's x and dz
        wv = w.clone().requires_grad_(True)
        F.conv2d(x, wv, None, stride=c.stride, padding=pad).backward(dz)
        assert _l2(L.dW.permute(0, 3, 1, 2).cpu(), wv.grad) < 2e-3, (L.name, "dW")
    # accumulated activation gradients: d(T) = sum over consumers of conv^T(dz) (+ d(out) for skip connections)
    seen = set()
    for L in P.layers:
        if L.frozen or L.dx is None or id(L.dx.buf) in seen:
            continue
        seen.add(id(L.dx.buf))
        full = torch.zeros(L.x.buf.shape[0], L.x.buf.shape[3], L.x.H, L.x.W)
        for M, kind in cons[id(L.x.buf)]:
            if M.frozen:
                continue
            if kind == "skip":
                full[:, M.residual.ch_offset:M.residual.ch_offset + M.residual.C] += M.dy.nchw().float().cpu()
                continue
            c = M.conv
            xin = M.x.nchw().float().cpu().requires_grad_(True)
            wq = q(c.weight.detach().float().cpu())
            dzm = M.dz.nchw()[:, :c.out_channels].float().cpu()
            F.conv2d(xin, wq, None, stride=c.stride, padding=(c.kernel_size - 1) // 2).backward(dzm)
            full[:, M.x.ch_offset:M.x.ch_offset + M.x.C] += xin.grad
        gbuf = L.dx.buf[:, L.x.halo:L.x.halo + L.x.H, L.x.halo:L.x.halo + L.x.W].permute(0, 3, 1, 2).float().cpu()
        # channels of a concat buffer that come from the upsample path are checked through their 2x2 sums below
        assert _l2(gbuf, full) < tol / 2, (L.name, "activation gradient", _l2(gbuf, full))
        checked_dx += 1
    for L in P.layers:
        if L.up_src is not None:
            ref = F.avg_pool2d(L.up_src.nchw().float().cpu(), 2) * 4
            assert float((L.dy.nchw().float().cpu() - ref).abs().max() / ref.abs().max()) < tol, (L.name, "sum2x2")
    assert checked_dx > 40
    model.eval()


def test_sgd_steps_reduce_the_loss(cuda):
    """A few optimizer steps on one fixed batch: the summed loss must go down (the gradients point downhill)."""
    model, cfg, sd, imgs, gb, gc, inputs = _make("bf16")
    model.train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
    hist = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        losses = model(inputs)
        total = sum(losses.values())
        total.backward()
        opt.step()
        hist.append(float(total.detach()))
    assert all(torch.isfinite(torch.tensor(hist)))
    assert hist[-1] < hist[0], hist
    model.eval()


def test_param_grads_alias_the_arena_and_still_accumulate(cuda):
    """The trainable parameters live in the engine's flat arena (channels_last conv weights) and ``.grad`` is a
    slice of the gradient arena; backward() twice without zero_grad() accumulates, a gradient tensor assigned by
    the caller is added to by autograd, and the state dict still has the reference's names, shapes and values."""
    model, cfg, sd, imgs, gb, gc, inputs = _make("f32", N=2, S=64)
    model.train()
    sum(model(inputs).values()).backward()
    eng = model.backbone.train_engine()
    views = eng.grad_views()
    assert len(views) == 210
    lo, hi = eng.grad_arena.data_ptr(), eng.grad_arena.data_ptr() + eng.grad_arena.numel() * 4
    for p, g in views:
        assert p.grad is not None and p.grad.data_ptr() == g.data_ptr() and lo <= g.data_ptr() < hi
        assert p.grad.shape == p.shape and p.grad.stride() == p.stride()
        assert eng.param_arena.data_ptr() <= p.data_ptr() < eng.param_arena.data_ptr() + eng.param_arena.numel() * 4
    w = model.backbone.bottom_up.res3[0].conv2.weight
    assert w.shape == (256, 128, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()
    for k, v in model.state_dict().items():   # adoption moved storage, not values
        if "running" not in k and "num_batches" not in k:
            assert v.shape == sd[k].shape and torch.equal(v.cpu(), sd[k]), k
    once = {id(p): p.grad.clone() for p, _ in views}
    # BatchNorm running statistics moved, the parameters did not: the second backward gives the same gradient
    sum(model(inputs).values()).backward()
    for p, _ in views[:40]:
        assert _l2(p.grad, 2 * once[id(p)]) < 1e-5
    # a gradient tensor of the caller's own: autograd adds to it, the arena slice is not installed
    for p, _ in views:
        p.grad = None
    p0, g0 = views[3]
    mine = torch.ones_like(p0)
    p0.grad = mine
    sum(model(inputs).values()).backward()
    assert p0.grad.data_ptr() == mine.data_ptr() and _l2(p0.grad - 1, once[id(p0)]) < 1e-5
    assert views[4][0].grad.data_ptr() == views[4][1].data_ptr()
    # storage replaced from outside is re-adopted on the next step
    with torch.no_grad():
        w.data = (w.data * 1.0).contiguous()
    for p, _ in views:
        p.grad = None
    sum(model(inputs).values()).backward()
    assert eng.param_arena.data_ptr() <= w.data_ptr() < eng.param_arena.data_ptr() + eng.param_arena.numel() * 4
    assert _l2(w.grad, once[id(w)]) < 1e-5
    model.eval()


def _torch_losses_on_heads(model, heads_nchw, gt_instances):
    """The torch-op path of the model (head activations, transposes, label assignment, losses) on given heads."""
    feats = [h.detach().clone().requires_grad_(True) for h in heads_nchw]
    losses = model.forward_training(None, feats, model.head(feats), gt_instances)
    sum(v * w for v, w in zip(losses.values(), (1.0, 0.5, 2.0))).backward()
    return losses, [f.grad for f in feats]


@pytest.mark.parametrize("case", ["random", "empty_image", "shared_anchor"])
def test_fused_loss_kernels_match_the_torch_ops(cuda, case):
    """yolo_train_loss_fwd / _bwd against the model's torch-op restatement of the reference's head + label
    assignment + losses (itself pinned to the oracle on the CPU in tests/test_train_host_cpu.py): the three losses,
    the positive count and every element of the head gradients, with unequal upstream weights."""
    from yolo_amd import ops
    from yolo_amd.structures import Boxes, Instances
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="f32")
    C, A, S, N = cfg.MODEL.YOLOV3.NUM_CLASSES, model.num_anchors, 96, 4
    g = torch.Generator().manual_seed(21)
    gts = []
    for n in range(N):
        k = int(torch.randint(1, 7, (1,), generator=g))
        if case == "empty_image" and n == 1:
            k = 0
        wh = 6 + torch.rand(k, 2, generator=g) * 70
        xy = torch.rand(k, 2, generator=g) * (S - wh).clamp_min(1)
        b = torch.cat([xy, xy + wh], 1)
        c = torch.randint(0, C, (k,), generator=g)
        if case == "shared_anchor" and n == 0:   # two identical boxes with different classes pick the same anchor
            b = torch.cat([b, b[:1]])
            c = torch.cat([c, (c[:1] + 1) % C])
        gts.append(Instances((S, S), gt_boxes=Boxes(b.to(cuda)), gt_classes=c.to(cuda)))
    strides = model._get_strides()
    ak = A * (5 + C)
    heads = [ops.alloc_view(N, S // s, S // s, (ak + 7) // 8 * 8, 0, torch.float32, cuda) for s in strides]
    for h in heads:
        h.buf.normal_(0, 1.5, generator=None)
    hv = [ops.View(h.buf, h.H, h.W, 0, ak, 0) for h in heads]
    want, dwant = _torch_losses_on_heads(model, [h.nchw() for h in hv], gts)
    boxes, classes, counts = model._padded_targets(gts)
    anchors = Boxes.cat(model.anchor_generator.grid_anchors([(h.H, h.W) for h in heads])).tensor.to(cuda).contiguous()
    losses, ctx = ops.train_loss_fwd(hv, strides, anchors, boxes, classes.to(torch.int32), counts, num_anchors=A,
                                     num_classes=C, match_thr=float(model.matcher_threshold),
                                     obj_weight=float(model.obj_weight), noobj_weight=float(model.noobj_weight))
    torch.cuda.synchronize()
    assert int(ctx.counts[0]) == int(model.num_pos_anchors)
    for got, (name, ref) in zip(losses.tolist(), want.items()):
        ref = float(ref.detach())
        assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref)), (name, got, ref)
    for dt in (torch.float32, torch.bfloat16):
        cpad = (ak + 63) // 64 * 64
        dheads = [ops.alloc_view(N, h.H, h.W, cpad, 1, dt, cuda) for h in heads]
        for d in dheads:
            d.buf.fill_(7.0)   # stale contents everywhere, halo included
        ops.train_loss_bwd(ctx, dheads, torch.tensor([1.0, 0.5, 2.0], device=cuda))
        torch.cuda.synchronize()
        for d, ref in zip(dheads, dwant):
            got = d.interior().float()
            ref = ref.permute(0, 2, 3, 1)
            tol = 1e-6 if dt == torch.float32 else 4e-3
            assert float((got[..., :ak] - ref).abs().max()) <= tol * float(ref.abs().max()), dt
            assert float(got[..., ak:].abs().max()) == 0.0
            assert float(d.buf[:, 0].min()) == 7.0   # the halo is not written


def test_padded_targets_from_host_and_device_instances_agree(cuda):
    """A loader hands over host tensors (the reference moves them with Instances.to(device), yolo.py:105), the
    synthetic benchmarks device tensors: both routes give the same padded [N,K,4] boxes, [N,K] classes and counts,
    an image without ground truth included; a batch without any box gives K = 1 rows of zeros."""
    from yolo_amd.structures import Boxes, Instances
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="f32")
    g = torch.Generator().manual_seed(3)
    host = []
    for k in (3, 0, 7, 1):
        b = torch.rand(k, 4, generator=g) * 90
        host.append(Instances((96, 96), gt_boxes=Boxes(b), gt_classes=torch.randint(0, 20, (k,), generator=g)))
    dev = [i.to(cuda) for i in host]
    bh, ch, nh = model._padded_targets(host)
    bd, cd, nd = model._padded_targets(dev)
    torch.cuda.synchronize()
    assert bh.device.type == "cuda" and bh.shape == (4, 7, 4) and ch.shape == (4, 7) and nh.tolist() == [3, 0, 7, 1]
    assert torch.equal(bh, bd) and torch.equal(ch, cd) and torch.equal(nh, nd)
    for n, inst in enumerate(host):
        k = len(inst)
        assert torch.equal(bh[n, :k].cpu(), inst.gt_boxes.tensor) and float(bh[n, k:].abs().sum()) == 0.0
        assert torch.equal(ch[n, :k].cpu(), inst.gt_classes)
    empty = [Instances((96, 96), gt_boxes=Boxes(torch.zeros(0, 4)), gt_classes=torch.zeros(0, dtype=torch.int64))] * 2
    b0, c0, n0 = model._padded_targets(empty)
    assert b0.shape == (2, 1, 4) and n0.tolist() == [0, 0] and float(b0.abs().sum()) == 0.0


def test_fused_step_equals_the_torch_op_step(cuda):
    """The model's default training step (fused loss) against the same model with fused_loss off: same losses and
    the same parameter gradients (f32 plan, so only the loss arithmetic differs)."""
    model, cfg, sd, imgs, gb, gc, inputs = _make("f32", N=3, S=96)
    model.train()
    eng = model.backbone.train_engine()
    res = {}
    for fused in (False, True):
        model.fused_loss = fused
        for p in model.parameters():
            p.grad = None
        model.load_state_dict(sd)   # same running statistics for both runs
        losses = model(inputs)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        res[fused] = ({k: float(v) for k, v in losses.items()},
                      {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    for k in res[True][0]:
        assert abs(res[True][0][k] - res[False][0][k]) <= 1e-5 * max(1.0, abs(res[False][0][k])), k
    assert len(res[True][1]) == 210 == len(res[False][1])
    for n, gref in res[False][1].items():
        assert _l2(res[True][1][n], gref) < 1e-4, n
    model.fused_loss = True
    model.eval()


def test_fused_loss_kernels_match_the_oracle_golden(cuda):
    """The HIP loss kernels against the committed vectors of the training oracle
    (tests/golden/oracle_train_golden.npz: seeded heads, ground truths incl. two sharing one anchor)."""
    import os
    import numpy as np
    from yolo_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_train_golden.npz"))
    C, A, strides, N = 20, 3, [8, 16, 32], 3
    ak = A * (5 + C)
    heads = []
    for i in range(3):
        h = torch.from_numpy(g[f"head{i}"]).to(cuda)
        v = ops.alloc_view(N, h.shape[2], h.shape[3], (ak + 7) // 8 * 8, 0, torch.float32, cuda)
        v.interior()[..., :ak] = h.permute(0, 2, 3, 1)
        heads.append(ops.View(v.buf, v.H, v.W, 0, ak, 0))
    gb = [torch.from_numpy(g[f"gt_boxes{n}"]) for n in range(N)]
    gc = [torch.from_numpy(g[f"gt_classes{n}"]) for n in range(N)]
    K = max(len(b) for b in gb)
    boxes, classes = torch.zeros(N, K, 4), torch.zeros(N, K, dtype=torch.int32)
    for n in range(N):
        boxes[n, :len(gb[n])], classes[n, :len(gb[n])] = gb[n], gc[n].to(torch.int32)
    counts = torch.tensor([len(b) for b in gb], dtype=torch.int32)
    from oracle import yolo_ref as R
    anchors = torch.cat(R.grid_anchors([(h.H, h.W) for h in heads], strides)).to(cuda).contiguous()
    losses, ctx = ops.train_loss_fwd(heads, strides, anchors, boxes.to(cuda), classes.to(cuda), counts.to(cuda),
                                     num_anchors=A, num_classes=C, match_thr=0.5, obj_weight=1.0, noobj_weight=100.0)
    lab = g["labels"]
    assert ctx.counts.tolist() == [int(((lab >= 0) & (lab < C)).sum()), int((lab == C).sum())]
    assert np.allclose(losses.cpu().numpy(), g["losses"], rtol=2e-6)
    dheads = [ops.alloc_view(N, h.H, h.W, 128, 1, torch.float32, cuda) for h in heads]
    ops.train_loss_bwd(ctx, dheads, torch.tensor([1.0, 0.5, 2.0], device=cuda))
    for i, d in enumerate(dheads):
        ref = torch.from_numpy(g[f"dhead{i}"]).permute(0, 2, 3, 1)
        assert float((d.interior()[..., :ak].cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_training_step_with_deeper_freeze_matches_oracle(cuda):
    """MODEL.BACKBONE.FREEZE_AT = 3 (stem, res1 and res2 frozen: the frozen stage beyond the fused prefix runs as
    folded inference convolutions inside the training plan): losses and gradients against the oracle, and no
    gradient, no running-statistics update in the frozen stages."""
    from oracle import yolo_train_ref as T
    from yolo_amd.structures import Boxes, Instances
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="f32", opts=["MODEL.BACKBONE.FREEZE_AT", 3])
    C, N, S = cfg.MODEL.YOLOV3.NUM_CLASSES, 2, 96
    imgs = workload.synthetic_images(N, S, S, seed=5)
    gb, gc = T.synthetic_targets(N, S, C, seed=13, max_boxes=5)
    inputs = [dict(image=im, instances=Instances((S, S), gt_boxes=Boxes(b), gt_classes=c)) for im, b, c in zip(imgs, gb, gc)]
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    full = dict(sd)
    full.update(params)
    want = T.forward_training(imgs.float() / 255.0, full, gb, gc, C, freeze_at=3)
    sum(want.values()).backward()
    model.train()
    got = model(inputs)
    sum(got.values()).backward()
    torch.cuda.synchronize()
    for k in want:
        assert abs(float(got[k].detach()) - float(want[k].detach())) / abs(float(want[k].detach())) < 1e-3, k
    frozen = set(T.frozen_names(3))
    assert any("res2" in f for f in frozen)
    n = 0
    for name, p in model.named_parameters():
        layer = name.rsplit(".norm.", 1)[0].rsplit(".weight", 1)[0].rsplit(".bias", 1)[0]
        if layer in frozen:
            assert p.grad is None and not p.requires_grad, name
            continue
        g, ref = p.grad.float().cpu(), params[name].grad
        assert float(F.cosine_similarity(g.flatten().double(), ref.flatten().double(), dim=0)) > 0.998, name
        assert _l2(g, ref) < 0.06, (name, _l2(g, ref))
        n += 1
    assert 0 < n < 210
    bufs = dict(model.named_buffers())
    key = "backbone.bottom_up.res2.1.conv2.norm.running_var"
    assert torch.equal(bufs[key].cpu(), sd[key])
    model.eval()


def test_inference_after_training_uses_the_updated_weights(cuda):
    """train() -> a few optimizer steps -> eval(): the inference plan repacks the (now channels_last, arena-resident)
    parameters and the updated running statistics, and its heads match the oracle on the current state dict."""
    from oracle import yolo_ref as R
    model, cfg, sd, imgs, gb, gc, inputs = _make("f32", N=2, S=64)
    model.eval()
    x = (imgs.float() / 255.0)
    before = {k: v.clone() for k, v in model.backbone(x.to(cuda)).items()}
    model.train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        sum(model(inputs).values()).backward()
        opt.step()
    model.eval()
    now = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    assert not torch.equal(now["backbone.fpn_output2.head.weight"], sd["backbone.fpn_output2.head.weight"])
    with torch.no_grad():
        ref = R.fpn_forward(x, now)
    out = model.backbone(x.to(cuda))
    torch.cuda.synchronize()
    for k in ("p3", "p4", "p5"):
        got = out[k].float().cpu()
        assert float((got - ref[k]).abs().max() / ref[k].abs().max()) < 1e-3, k
        assert not torch.allclose(got, before[k].float().cpu())


def test_fused_sgd_step_equals_torch_sgd(cuda, monkeypatch):
    """ArenaSGD's one-launch step against torch.optim.SGD's own step (YOLO_FUSED_SGD=0) from the same start over the
    same batches: parameters and momentum buffers agree to rounding (torch may contract a + alpha * b into one
    rounding, the kernel never does: 1e-6 relative), version counters move, the state dict keeps torch's layout."""
    import workload
    from yolo_amd.trainer import Trainer, synthetic_batches

    def run(fused):
        monkeypatch.setenv("YOLO_FUSED_SGD", "1" if fused else "0")
        torch.manual_seed(0)
        model, cfg = workload.build_model(config="yolov3.yaml", dtype="bf16",
                                          opts=["SOLVER.IMS_PER_BATCH", 4, "SOLVER.BASE_LR", 1e-4, "SOLVER.WARMUP_ITERS", 0])
        data = synthetic_batches(cfg, 4, 128, seed=5, device=model.device)
        tr = Trainer(cfg, model, data, log=lambda m: None)
        v0 = model.backbone.fpn_output0.head.weight._version
        tr.run_step()  # compared after ONE step: later ones amplify the last-bit differences through the network
        used = tr.optimizer._fused is not None
        params = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
        moms = [tr.optimizer.state[p]["momentum_buffer"].detach().float().cpu().clone()
                for g in tr.optimizer.param_groups for p in g["params"]]
        for _ in range(2):
            losses = tr.run_step()
            assert all(bool(torch.isfinite(v)) for v in losses.values())
        assert (tr.optimizer._fused is not None) == used
        assert model.backbone.fpn_output0.head.weight._version >= v0 + 3
        sd = tr.optimizer.state_dict()
        assert len(sd["state"]) == 210 and all("momentum_buffer" in s for s in sd["state"].values())
        return used, params, moms

    used_f, pf, mf = run(True)
    used_t, pt, mt = run(False)
    assert used_f and not used_t
    for n in pf:
        assert torch.allclose(pf[n], pt[n], rtol=1e-5, atol=1e-7), n
    for a, b in zip(mf, mt):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_side_stream_weight_gradients_equal_the_single_stream_ones(cuda):
    """Backward with the weight gradients on their own stream (the default) gives bit for bit the gradients of
    backward on one stream: same kernels, the streams only reorder independent launches -- a missing wait between
    them would show up here."""
    model, cfg, sd, imgs, gb, gc, inputs = _make("bf16", N=4, S=160)
    model.train()
    eng = model.backbone.train_engine()
    grads = {}
    for side in (True, False, True):
        eng.wgrad_on_side_stream = side
        model.zero_grad(set_to_none=True)
        losses = model(inputs)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        got = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        assert len(got) == 210
        if not grads:
            grads = got
            continue
        for n in grads:
            assert torch.equal(got[n], grads[n]), (side, n)


def test_bn_backward_sums_from_the_dgrad_epilogue_give_the_same_gradients(cuda):
    """The plan that takes the BatchNorm-backward reduction from the data-gradient epilogue (the default: 63 of the 68
    BatchNorm layers) against the plan that runs the reduction as a pass of its own.  The sums are formed from the same
    stored dy and z in another order, so a layer's dz moves by a bf16 rounding in a few elements -- and every layer
    further down the backward pass sees those and adds its own: the head convolutions (nothing fused behind them) agree
    bit for bit, the first fused layer the backward pass meets within 1e-3, and from there the relative L2 difference of the
    parameter gradients grows layer by layer to 1-2e-2 at the first trainable stage (measured: 4e-5, 2e-4, 9e-4, 2e-3,
    3e-3 ... 2e-2), the level at which two bf16 evaluation orders of this freshly initialised network differ anyway."""
    model, cfg, sd, imgs, gb, gc, inputs = _make("bf16", N=4, S=160)
    model.train()
    eng = model.backbone.train_engine()
    grads = {}
    fused_layers = 0
    for fused in (False, True):
        eng.fuse_bn_bwd_sums = fused
        eng._plans.clear()
        model.zero_grad(set_to_none=True)
        losses = model(inputs)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        if fused:
            fused_layers = sum(L.sums_for is not None for L in next(iter(eng._plans.values())).layers)
        grads[fused] = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
    eng.fuse_bn_bwd_sums = True
    assert fused_layers >= 60, fused_layers
    err = {n: _l2(grads[True][n], grads[False][n]) for n in grads[False]}
    for n, e in err.items():
        if ".head." in n:
            assert e == 0.0, (n, e)
        if "fpn_output" in n and ".conv." in n:   # its dy comes from the head's data gradient: only the sums' order differs
            assert e < 1e-5, (n, e)
        if "fpn_lateral2.blocks.4." in n:   # the finest level: the first fused layer the backward pass meets
            assert e < 1e-3, (n, e)
    worst = max(err.values())
    print("relative L2 difference of the parameter gradients: worst %.2e" % worst)
    assert worst < 5e-2, max((e, n) for n, e in err.items())
    model.eval()


def test_multi_scale_training_80_classes(cuda):
    """configs/yolov3-aug.yaml (80 classes) with the side of the batch drawn per step from the multi-scale list
    (the reference's INPUT.MIN_SIZE_TRAIN, configs/yolov3.yaml:36; yolov3-aug.yaml:35-38 trains the same network):
    six steps over {320, 416, 512, 608} with one plan per side, finite losses throughout, the resident plans bounded by
    the cache's limits (least recently used plan evicted), and the first step at the smallest side equal to the
    oracle's training forward on the same state dict (fp32 plan)."""
    from oracle import yolo_train_ref as T
    from yolo_amd.structures import Boxes, Instances
    model, cfg = workload.build_model(config="yolov3-aug.yaml", dtype="f32")
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    assert C == 80
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-5, momentum=0.9)
    eng = model.backbone.train_engine()
    N = 2
    sides = [320, 608, 416, 512, 320, 608]
    one_plan = None
    for step, S in enumerate(sides):
        imgs = workload.synthetic_images(N, S, S, seed=50 + step)
        gb, gc = T.synthetic_targets(N, S, C, seed=60 + step, max_boxes=8)
        inputs = [dict(image=im, instances=Instances((S, S), gt_boxes=Boxes(b), gt_classes=c))
                  for im, b, c in zip(imgs, gb, gc)]
        if step == 0:
            with torch.no_grad():
                want = T.forward_training(imgs.float() / 255.0, sd, gb, gc, C, freeze_at=2)
        opt.zero_grad(set_to_none=False)
        losses = model(inputs)
        total = sum(losses.values())
        total.backward()
        assert all(bool(torch.isfinite(v)) for v in losses.values()), (step, S, losses)
        if step == 0:
            for k in want:
                assert abs(float(losses[k]) - float(want[k])) / abs(float(want[k])) < 1e-3, (k, float(losses[k]), float(want[k]))
            assert eng._plans.nbytes > 0
            eng._plans.max_plans = 2  # from here on at most two sides stay resident
        opt.step()
        assert len(eng._plans) <= 2
    torch.cuda.synchronize()
    assert eng._plans.evictions >= 3 and (2, 608, 608) in eng._plans
    # the byte budget evicts as well: with a budget below one plan only the newest plan stays
    eng._plans.budget_bytes = 1
    # an evicted side is rebuilt on demand and still trains
    S = 416
    imgs = workload.synthetic_images(N, S, S, seed=99)
    gb, gc = T.synthetic_targets(N, S, C, seed=98, max_boxes=8)
    losses = model([dict(image=im, instances=Instances((S, S), gt_boxes=Boxes(b), gt_classes=c)) for im, b, c in zip(imgs, gb, gc)])
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    assert eng._plans.keys() == [(2, 416, 416)]
    model.eval()


def test_training_trajectory_fp32_follows_the_oracle(cuda):
    """Four whole iterations of the loop -- forward, backward, SGD with momentum and the two weight-decay groups, the
    warm-up schedule, BatchNorm running statistics carried from step to step -- next to the oracle's autograd driven by
    torch.optim.SGD on the CPU.  Training this randomly initialised network is CHAOTIC: the oracle run in float64
    instead of float32, or with its weights perturbed by 1e-6, already differs by 30 % in the regression loss of the
    third step, and its four-step parameter update has a cosine of only 0.77-0.79 with the unperturbed one (norm ratio
    1.01).  What a step hands to the next is therefore checked on the quantities that stay smooth: the first step's
    losses exactly (1e-4: nothing has diverged yet), the second step's within 6 % (one update with the warm-up's first
    learning rate lies between them; a 1e-6 perturbation of the oracle moves its regression loss there by 4 %), the dominant objectness loss of every step within 3 %, the NORM of the
    four-step update within 5 % of the oracle's -- a missing warm-up, a wrong momentum or a decayed normalisation
    group change it by tens of per cent -- with a cosine above 0.6, and the running statistics (5e-3 for the first
    trainable layer, 10 % deep in the network)."""
    from oracle import yolo_train_ref as T
    from yolo_amd.structures import Boxes, Instances
    from yolo_amd.trainer import Trainer
    N, S, base_lr, warm_iters, warm_factor, steps = 4, 160, 0.0005, 2, 0.5, 4
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="f32",
                                      opts=["OUTPUT_DIR", "", "SOLVER.CHECKPOINT_PERIOD", 0, "SOLVER.BASE_LR", base_lr,
                                            "SOLVER.WARMUP_ITERS", warm_iters, "SOLVER.WARMUP_FACTOR", warm_factor])
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    imgs = workload.synthetic_images(N, S, S, seed=3)
    gb, gc = T.synthetic_targets(N, S, C, seed=11, max_boxes=5)
    inputs = [dict(image=im, instances=Instances((S, S), gt_boxes=Boxes(b), gt_classes=c)) for im, b, c in zip(imgs, gb, gc)]
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # ---- oracle: autograd + torch.optim.SGD with the reference's groups (detectron2 build_optimizer: no decay on norms)
    frozen = set(T.frozen_names(2))
    layer_of = lambda k: k.rsplit(".norm.", 1)[0].rsplit(".weight", 1)[0].rsplit(".bias", 1)[0]  # noqa: E731
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and "running" not in k and layer_of(k) not in frozen}
    opt = torch.optim.SGD([dict(params=[p for k, p in params.items() if ".norm." not in k], weight_decay=cfg.SOLVER.WEIGHT_DECAY),
                           dict(params=[p for k, p in params.items() if ".norm." in k], weight_decay=cfg.SOLVER.WEIGHT_DECAY_NORM)],
                          lr=base_lr, momentum=cfg.SOLVER.MOMENTUM)
    state = dict(sd)
    want = []
    for it in range(steps):
        alpha = min(it / warm_iters, 1.0)
        for g in opt.param_groups:
            g["lr"] = base_lr * (warm_factor * (1 - alpha) + alpha if it < warm_iters else 1.0)
        full = dict(state)
        full.update(params)
        stats = {}
        losses = T.forward_training(imgs.float() / 255.0, full, gb, gc, C, freeze_at=2, stats=stats)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        for name, (rm, rv) in stats.items():
            state[name + ".norm.running_mean"], state[name + ".norm.running_var"] = rm, rv
        want.append({k: float(v.detach()) for k, v in losses.items()})
    # ---- the HIP path, through the Trainer
    tr = Trainer(cfg, model, iter(lambda: inputs, None), log=lambda m: None)
    got = [{k: float(v) for k, v in tr.run_step().items()} for _ in range(steps)]
    torch.cuda.synchronize()
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-6)  # noqa: E731
    for k in want[0]:
        assert rel(got[0][k], want[0][k]) < 1e-4, (0, k, got[0][k], want[0][k])
        assert rel(got[1][k], want[1][k]) < 6e-2, (1, k, got[1][k], want[1][k])
    for it in range(steps):
        assert rel(got[it]["confs_loss"], want[it]["confs_loss"]) < 3e-2, (it, got[it], want[it])
    assert want[-1]["confs_loss"] < 0.7 * want[0]["confs_loss"]   # the steps did move the model
    live = dict(model.named_parameters())
    for sel in (lambda k: True, lambda k: ".norm." in k, lambda k: ".norm." not in k):
        d_got = torch.cat([(live[k].detach().float().cpu() - sd[k]).flatten() for k in params if sel(k)]).double()
        d_ref = torch.cat([(p.detach() - sd[k]).flatten() for k, p in params.items() if sel(k)]).double()
        cos = float(F.cosine_similarity(d_got, d_ref, dim=0))
        ratio = float(d_got.norm() / d_ref.norm())
        assert cos > 0.6 and 0.95 < ratio < 1.05, (cos, ratio)
    bufs = dict(model.named_buffers())
    # the first trainable layer sees the frozen prefix's output: its statistics only feel its own weights' drift; deep
    # layers feel everyone's
    for name, tol in (("backbone.bottom_up.res2.0.downsample", 5e-3), ("backbone.bottom_up.res3.2.conv2", 0.1),
                      ("backbone.fpn_lateral0.blocks.0", 0.1), ("backbone.fpn_output2.conv", 0.1)):
        var = state[name + ".norm.running_var"]
        assert _l2(bufs[name + ".norm.running_var"].cpu(), var) < tol, name
        # means are differences of large terms: measured against the channel's standard deviation
        dm = (bufs[name + ".norm.running_mean"].cpu() - state[name + ".norm.running_mean"]).abs() / var.sqrt()
        assert float(dm.max()) < 4 * tol, (name, float(dm.max()))


def test_training_fits_a_structured_set_and_the_inference_path_finds_it(cuda):
    """The whole loop as a user runs it (reference tools/train_net.py -> DefaultTrainer.run_step, then the model in eval
    mode): 400 SGD steps in bf16 on 32 synthetic images with real structure (filled rectangles, class = colour) must
    bring the summed loss down by more than 100 x, and the trained model's own inference path (running BatchNorm
    statistics, decode, NMS, top-100) must then find at least 80 % of the training rectangles with the right class at
    IoU > 0.5 -- none before training.  (Measured: loss 114 -> 0.013, recall 0 -> 0.97 after 600 steps.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import overfit_check
    res = overfit_check.run(steps=400, n_images=32, side=256, dtype="bf16", log=lambda m: None)
    assert res["recall_before"] <= 0.1
    assert res["loss_last"] < res["loss_first"] / 100, res
    assert res["recall_after"] >= 0.8, res
    assert res["detections_per_image_after"] < 20, res


def test_bf16_gradients_match_the_fp32_oracle_on_trained_weights(cuda):
    """The bf16 training plan against the fp32 oracle's autograd DIRECTLY -- possible once the weights are trained (on
    the randomly initialised network a bf16 forward already differs by 1-25 % at the heads, which is why the other
    bf16 tests compare with torch autograd on the plan's own tensors): after 400 steps on the structured synthetic
    set, one forward / backward on fresh images of the same kind gives losses within 1e-2 of the oracle's and, for
    each of the 210 trainable tensors, a gradient with cosine > 0.95 to the oracle's (measured: minimum 0.988, median
    0.9965; all gradients as one vector: cosine 0.996, norm ratio 1.001)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import trained_bf16_grads
    out = trained_bf16_grads.compare(steps=400, log=lambda m: None)
    for k, v in out["losses_oracle"].items():
        assert abs(out["losses_hip"][k] - v) <= 1e-2 * abs(v), (k, out)
    assert out["tensors"] == 210
    assert out["cos_min"] > 0.95 and out["cos_all"] > 0.99 and 0.98 < out["norm_ratio_all"] < 1.02, out
