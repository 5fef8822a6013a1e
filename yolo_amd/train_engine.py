"""Training step of Darknet-53 + FPN as an explicit forward / backward plan of gfx950 kernel launches.

What the reference runs through torch autograd (tools/train_net.py -> detectron2 DefaultTrainer:
``losses = model(data); losses.backward()`` over yolo/modeling/backbone/darknet.py:113-126,203-226 and
fpn.py:33-37,66-71,96-100,165-185 with the modules in training mode) is laid out here once per batch shape:

  forward, per trainable conv:   yolo_conv2d_fwd (raw weights, linear)  ->  z
                                 yolo_bn_train_stats(z) (+ running-stat update)
                                 yolo_bn_act_fwd(z) (+ residual)         ->  y      [heads: conv + bias only]
  backward, in reverse order:    yolo_bn_act_bwd(dy, z)                  ->  dz, dgamma, dbeta
                                 (its reduction pass comes out of the epilogue of the launch that wrote dy where
                                 that is the layer behind it: yolo_conv2d_fwd_bnbwd_sums + yolo_bn_act_bwd_presummed)
                                 yolo_conv2d_wgrad(x, dz)                ->  dW
                                 yolo_conv2d_fwd(dz [dilated for stride 2], W^T rot180) (+ other consumers'
                                 gradient as the fused residual term)    ->  dx

Frozen stages (cfg.MODEL.BACKBONE.FREEZE_AT, darknet.py:236-258: FrozenBatchNorm, no gradients) run as the
fused inference kernels and take no part in the backward pass.  Activations are bf16 NHWC with a zero halo;
every activation that backward needs has its own buffer (nothing is recomputed); weight, gamma and beta
gradients are fp32.  nn.Upsample + torch.cat stay free in forward (channel slices of one buffer); their
backward is a channel slice of the gradient buffer plus one 2x2-sum kernel.

Parameters and their gradients live in two flat fp32 arenas with one layout, in the order backward produces the
gradients: conv weights pixel-major [Cout][kh][kw][Cin] (to torch: channels_last tensors of the usual
[Cout,Cin,kh,kw] shape), then gamma and beta (or the head's bias).  The trainable nn.Parameters are re-pointed at
views of the parameter arena the first time a plan is built ("adopted"); the weight-gradient kernel writes its
result in that same layout, so ``param.grad`` can alias the gradient arena -- no per-parameter copies, the
optimizer's multi-tensor kernels see matching strides, and data-parallel training all-reduces arena ranges.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import View
from .utils.plan_cache import PlanCache

LEAKY = 0.1
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclass
class TLayer:
    name: str
    conv: object
    x: View
    y: View                      # post-activation output (head: the f32 head tensor)
    z: Optional[View] = None     # conv output before BatchNorm (trainable BN layers)
    residual: Optional[View] = None
    frozen: bool = False
    stem: bool = False          # the 3-channel first convolution (reads the images, has its own kernels)
    slope: float = LEAKY
    up_dst: Optional[View] = None     # forward: y is also written 2x-upsampled here
    dy: Optional[View] = None         # gradient w.r.t. y
    dz: Optional[View] = None         # gradient w.r.t. z (heads: bf16 copy of the head gradient)
    dzd: Optional[View] = None        # stride 2: dz with zeros inserted
    dx: Optional[View] = None         # where the data gradient goes (None: not needed)
    dx_add: Optional[View] = None     # second term of dx (the skip connection's gradient)
    dx_accumulate: bool = False       # dx already holds another consumer's gradient
    up_src: Optional[View] = None     # backward: dy = 2x2 sums of this
    stats: Optional[torch.Tensor] = None
    dW: Optional[torch.Tensor] = None
    dgamma: Optional[torch.Tensor] = None
    dbeta: Optional[torch.Tensor] = None
    dbias: Optional[torch.Tensor] = None
    arena_end: int = 0                # end of this layer's gradients inside the plan's arena
    sums_for: Optional["TLayer"] = None   # this layer's data-gradient launch also reduces that layer's BatchNorm backward
    presummed_rows: int = 0           # rows of such sums waiting in the plan's bn_sums buffer (set during backward)


@dataclass
class TPlan:
    N: int
    H: int
    W: int
    layers: List[TLayer] = field(default_factory=list)
    heads: List[View] = field(default_factory=list)       # p3, p4, p5 (f32, 8-padded channels)
    dheads: List[View] = field(default_factory=list)      # bf16 gradient buffers of the heads
    prefix: List[Callable] = field(default_factory=list)  # frozen fused launches after the stem
    images: Optional[torch.Tensor] = None   # the step's input (a trainable stem's weight gradient reads it again)
    norm: Optional[tuple] = None             # its (mean, std)
    stem_y: Optional[View] = None
    stem_out: Optional[View] = None
    bn_ws: Optional[torch.Tensor] = None
    bn_sums: Optional[torch.Tensor] = None  # per-tile BatchNorm sums written by the conv epilogue
    wg_ws: Optional[torch.Tensor] = None
    arena: Optional[torch.Tensor] = None   # all parameter gradients, flat fp32, in backward order
    bucketer: Optional[object] = None


class TrainEngine:
    def __init__(self, backbone, *, num_classes: int, num_anchors: int, freeze_at: int = 2,
                 blocks: Sequence[int] = (1, 2, 8, 8, 4), dtype: torch.dtype = torch.bfloat16):
        self.backbone = backbone
        self.num_classes, self.num_anchors = num_classes, num_anchors
        self.blocks = list(blocks)
        self.freeze_at = freeze_at
        self.device = next(backbone.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("TrainEngine needs the model on an MI355X (cuda device); there is no CPU path")
        if freeze_at < 0:
            raise ValueError(f"MODEL.BACKBONE.FREEZE_AT = {freeze_at}: must be >= 0")
        self.dtype = dtype  # bf16: the fast path; f32: the 1e-3 parity plan (unfused prefix, FMA weight gradient)
        self.convs = dict(backbone.darknet_modules())
        self.frozen = set()
        names = ["bottom_up.stem.conv"] if freeze_at >= 1 else []
        for si, nb in enumerate(self.blocks, start=1):
            if freeze_at >= si + 1:
                for b in range(nb):
                    p = f"bottom_up.res{si}.{b}"
                    names += ([p + ".downsample"] if b == 0 else []) + [p + ".conv1", p + ".conv2"]
        self.frozen.update(names)
        self._plans = PlanCache(self.device)  # (N, H, W) -> TPlan; multi-scale training keeps one plan per scale
        self._wcache: Dict[str, tuple] = {}
        self._wbufs: Dict[str, tuple] = {}
        self._zero_bias: Dict[int, torch.Tensor] = {}
        self._wstream = None
        self.wgrad_on_side_stream = os.environ.get("YOLO_WGRAD_STREAM", "1") != "0"
        self.bucket_bytes = 64 << 20
        self.process_group = None
        self.fuse_bn_sums = os.environ.get("YOLO_FUSE_BN_SUMS", "1") != "0"
        self.fuse_bn_bwd_sums = os.environ.get("YOLO_FUSE_BN_BWD_SUMS", "1") != "0"
        self.param_arena: Optional[torch.Tensor] = None
        self.grad_arena: Optional[torch.Tensor] = None
        self._adopted: List[Tuple[torch.nn.Parameter, torch.Tensor, torch.Tensor]] = []  # (param, data view, grad view)
        self._slots: Dict[str, Tuple[int, int]] = {}     # layer name -> (offset of dW in the arenas, arena end)
        self._order: List[str] = []                       # trainable layers in backward order
        self._pack_tbl = None
        self._pack_ver = None

    # ------------------------------------------------------------------ packing (weights change every step)
    def _zeros(self, n: int) -> torch.Tensor:
        if n not in self._zero_bias:
            self._zero_bias[n] = torch.zeros(n, device=self.device)
        return self._zero_bias[n]

    def _train_bufs(self, name: str):
        """Operand buffers of a trainable conv, allocated and zero-padded once: (forward weights, data-gradient
        weights, forward bias, zero bias of the data-gradient conv).  "raw" = the bare convolution (+ bias for the
        heads); "dgrad" = Cin <-> Cout with the taps rotated, Cout padded to what the kernel reads."""
        bufs = self._wbufs.get(name)
        if bufs is None:
            conv = self.convs[name]
            k, cin, cout = conv.kernel_size, conv.in_channels, conv.out_channels
            cpad = (cout + 63) // 64 * 64
            mk = lambda rows, cols: torch.zeros(rows, cols, dtype=self.dtype, device=self.device)  # noqa: E731
            extra = 32 if (self.dtype == torch.bfloat16 and k == 3) else 0   # a 32-channel row spans two taps
            bufs = (mk(ops.lib().yolo_conv_cout_pad(cout), k * k * cin + (extra if cin == 32 else 0)),
                    mk(ops.lib().yolo_conv_cout_pad(cin), k * k * cpad + (extra if cpad == 32 else 0)),
                    torch.zeros(ops.lib().yolo_conv_cout_pad(cout), device=self.device),
                    torch.zeros(ops.lib().yolo_conv_cout_pad(cin), device=self.device))
            self._wbufs[name] = bufs
        return bufs

    def _pack_all(self) -> None:
        """Repack every trainable conv's master weights into its operand buffers -- one launch, and only when an
        optimizer step (or a checkpoint load) has touched them since the last time."""
        self._check_adopted()
        convs = [self.convs[n] for n in self._order]
        ver = tuple((c.weight._version, None if c.bias is None else c.bias._version) for c in convs)
        if ver == self._pack_ver:
            return
        if self._pack_tbl is None:
            entries = []
            for n, c in zip(self._order, convs):
                if c.in_channels == 3:
                    continue  # the stem has its own kernels and operand layout
                b = self._train_bufs(n)
                entries.append((c.weight.data, b[0], b[1], (c.out_channels + 63) // 64 * 64))
            self._pack_tbl = ops.pack_table(entries, self.device)
        ops.pack_conv_train_batch(*self._pack_tbl, self.dtype)
        for n, c in zip(self._order, convs):
            if c.in_channels == 3:    # [Cout, 3, kh, kw] -> [kh, kw, 3, Cout], what yolo_stem_fwd reads
                self._stem_w = c.weight.detach().permute(2, 3, 1, 0).contiguous().float()
            elif c.bias is not None:
                self._wbufs[n][2][:c.out_channels].copy_(c.bias.detach())
        self._pack_ver = ver

    # ------------------------------------------------------------------ flat parameter / gradient arenas
    def _adopt(self, order: "List[TLayer]") -> None:
        """Lay out the arenas for the trainable layers (in backward order) and re-point the parameters at them."""
        total = sum(L.conv.weight.numel() + (2 if L.conv.norm is not None else 1) * L.conv.out_channels for L in order)
        self.param_arena = torch.empty(total, device=self.device)
        self.grad_arena = torch.zeros(total, device=self.device)
        off = 0
        for L in order:
            c = L.conv
            co, ci, k = c.out_channels, c.in_channels, c.kernel_size
            start = off
            parts = [(c.weight, c.weight.numel(), lambda a: a.view(co, k, k, ci).permute(0, 3, 1, 2))]
            flat = lambda a: a  # noqa: E731
            if c.norm is not None:
                parts += [(c.norm.weight, co, flat), (c.norm.bias, co, flat)]
            else:
                parts += [(c.bias, co, flat)]
            for prm, n, shape in parts:
                self._adopted.append((prm, shape(self.param_arena[off:off + n]), shape(self.grad_arena[off:off + n])))
                off += n
            self._slots[L.name] = (start, off)
            self._order.append(L.name)
        self._check_adopted()

    def _check_adopted(self) -> None:
        """A parameter whose storage was replaced from outside (module.to(), an assigned state dict) is copied
        back into the arena and re-pointed; in-place updates (optimizer steps, load_state_dict) never trigger this."""
        for prm, view, _ in self._adopted:
            if prm.data_ptr() != view.data_ptr() or prm.stride() != view.stride():
                with torch.no_grad():
                    view.copy_(prm.data)
                prm.data = view
                self._pack_ver = None

    def grad_views(self) -> "List[Tuple[torch.nn.Parameter, torch.Tensor]]":
        """(parameter, its slice of the gradient arena shaped and strided like the parameter)."""
        return [(prm, g) for prm, _, g in self._adopted]

    def _packed(self, name: str, kind: str):
        if kind in ("raw", "dgrad") and name not in self.frozen:
            b = self._train_bufs(name)
            return (b[0], b[2]) if kind == "raw" else (b[1], b[3])
        conv = self.convs[name]
        ver = (conv.weight._version, None if conv.bias is None else conv.bias._version,
               None if conv.norm is None else tuple(t._version for t in conv.norm.as_dict().values()))
        key = (name, kind)
        hit = self._wcache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        if kind == "folded":       # frozen layer: eval BatchNorm folded in
            w, b = conv.folded()
            pw, pb = (ops.pack_stem(w, b, self.device) if conv.in_channels == 3
                      else ops.pack_conv(w, b, self.dtype, self.device))
        else:
            raise KeyError(f"{name}: trainable layers are packed by _pack_all()")
        self._wcache[key] = (ver, pw, pb)
        return pw, pb

    # ------------------------------------------------------------------ plan
    @ops.on_own_device
    def plan(self, N: int, H: int, W: int) -> TPlan:
        return self._plans.get((N, H, W), lambda: self._build(N, H, W))

    def _build(self, N: int, H: int, W: int) -> TPlan:
        assert H % 32 == 0 and W % 32 == 0
        dt, dev = self.dtype, self.device
        P = TPlan(N, H, W)
        A = lambda h, w, c, halo=1, dtype=dt: ops.alloc_view(N, h, w, c, halo, dtype, dev)  # noqa: E731
        written = set()  # ids of gradient buffers that already hold a consumer's contribution

        def add(name, x, y, *, residual=None, slope=LEAKY, dy=None, up_dst=None, up_src=None):
            conv = self.convs[name]
            L = TLayer(name, conv, x, y, residual=residual, slope=slope, frozen=name in self.frozen,
                       up_dst=up_dst, up_src=up_src, dy=dy)
            if not L.frozen:
                ho, wo = y.H, y.W
                if conv.norm is not None:
                    L.z = A(ho, wo, conv.out_channels, 0)
                    L.stats = torch.zeros(5, conv.out_channels, device=dev)
                    wide = A(ho, wo, (conv.out_channels + 63) // 64 * 64)   # zero-padded to what the data-gradient conv reads
                    L.dz = View(wide.buf, ho, wo, wide.halo, conv.out_channels, 0)
                if conv.stride == 2 and not (dt == torch.bfloat16 and conv.kernel_size == 3 and conv.in_channels > 64):
                    L.dzd = A(2 * ho, 2 * wo, conv.out_channels)   # zero-dilated dz for the generic data-gradient form
            P.layers.append(L)
            return L

        grads: Dict[int, View] = {}  # activation buffer id -> gradient view of the same geometry

        def grad_of(v: View) -> View:
            """Gradient view matching activation view ``v`` (same buffer geometry and channel slice)."""
            k = id(v.buf)
            if k not in grads:
                grads[k] = View(torch.zeros_like(v.buf), v.H, v.W, v.halo, v.buf.shape[3], 0)
            gb = grads[k]
            return View(gb.buf, v.H, v.W, v.halo, v.C, v.ch_offset)

        # ---- forward topology (mirrors yolo_amd/engine.py)
        stem = self.convs["bottom_up.stem.conv"]
        c1 = self.convs["bottom_up.res1.0.downsample"].out_channels
        chans = [c1 * 2 ** i for i in range(5)]
        cat = {3: A(H // 8, W // 8, chans[2] + chans[2] // 2), 4: A(H // 16, W // 16, chans[3] + chans[3] // 2)}
        h, w = H, W
        fused_prefix = (dt == torch.bfloat16 and stem.out_channels == 32 and c1 == 64 and self.blocks[0] >= 1
                        and self.freeze_at >= 2)
        x = None
        stage_out: Dict[int, View] = {}
        if not fused_prefix:
            P.stem_out = A(H, W, stem.out_channels)
            x = P.stem_out
            if "bottom_up.stem.conv" not in self.frozen:
                add("bottom_up.stem.conv", None, P.stem_out).stem = True
        for si, nb in enumerate(self.blocks, start=1):
            h, w = h // 2, w // 2
            co = chans[si - 1]
            if si == 1 and fused_prefix:
                X, X2 = A(h, w, co), A(h, w, co)
                P.stem_y = X
                for b in range(nb):
                    p = f"bottom_up.res1.{b}"
                    P.prefix.append(lambda p=p, a=X, o=X2: ops.resblock_c64_fwd(
                        a, *self._packed(p + ".conv1", "folded"), *self._packed(p + ".conv2", "folded"), o))
                    X, X2 = X2, X
            else:
                X = A(h, w, co)
                add(f"bottom_up.res{si}.0.downsample", x, X)
                for b in range(nb):
                    p = f"bottom_up.res{si}.{b}"
                    T = A(h, w, co // 2)
                    add(p + ".conv1", X, T)
                    last = b == nb - 1
                    Y = cat[si].slice(co // 2, co) if (last and si in cat) else A(h, w, co)
                    add(p + ".conv2", T, Y, residual=X)
                    X = Y
            stage_out[si] = X
            x = X
        ak = (5 + self.num_classes) * self.num_anchors
        ak_pad = (ak + 7) // 8 * 8
        feats = [stage_out[5], cat[4], cat[3]]
        heads, dheads = [], []
        for idx in range(3):
            f = feats[idx]
            lat = self.convs[f"fpn_lateral{idx}.blocks.0"].out_channels
            fh, fw = f.H, f.W
            src = f
            for i in range(5):
                dst = A(fh, fw, lat if i % 2 == 0 else lat * 2)
                add(f"fpn_lateral{idx}.blocks.{i}", src, dst)
                src = dst
            lat_out = src
            B = A(fh, fw, lat * 2)
            add(f"fpn_output{idx}.conv", lat_out, B)
            hv = A(fh, fw, ak_pad, 0, torch.float32)
            L = add(f"fpn_output{idx}.head", B, View(hv.buf, fh, fw, 0, ak_pad, 0), slope=1.0)
            L.dz = A(fh, fw, (ak + 63) // 64 * 64)  # bf16 copy of the head gradient, padded for the dgrad conv
            heads.insert(0, View(hv.buf, fh, fw, 0, ak, 0))
            dheads.insert(0, L.dz)
            if idx < 2:
                up_c = self.convs[f"fpn_upsample{idx}.conv"].out_channels
                t = A(fh, fw, up_c)
                add(f"fpn_upsample{idx}.conv", lat_out, t, up_dst=cat[4 - idx].slice(0, up_c))
        P.heads, P.dheads = heads, dheads

        # ---- backward wiring: walk the layers in reverse and decide where each data gradient lands
        first_trainable = next(i for i, L in enumerate(P.layers) if not L.frozen)
        for i in range(len(P.layers) - 1, -1, -1):
            L = P.layers[i]
            if L.frozen:
                continue
            if L.conv.norm is not None:
                if L.up_dst is not None:   # gradient arrives through the 2x upsample into the concat buffer
                    L.up_src = grad_of(L.up_dst)
                    L.dy = View(torch.zeros_like(L.y.buf), L.y.H, L.y.W, L.y.halo, L.y.C, 0)
                else:
                    L.dy = grad_of(L.y)
            if i == first_trainable or self._producer_frozen(P, L):
                continue                   # its input comes from frozen layers: no data gradient
            L.dx = grad_of(L.x)
            kx = id(L.dx.buf)
            if self._is_conv1(L):          # block input also feeds the skip connection: + d(block output)
                blk_out = P.layers[i + 1]
                assert blk_out.residual is not None and blk_out.residual.buf is L.x.buf
                assert kx not in written, "block input with a third consumer"
                L.dx_add = grad_of(blk_out.y)
            elif kx in written:
                L.dx_accumulate = True
            written.add(kx)
        # The reduction pass of a BatchNorm's backward (two reads of the activation-sized dy and z) comes out of the
        # epilogue of the data-gradient launch that writes dy, when that launch is the last one to write it: the
        # consumer directly behind the producer in the plan, reading exactly the producer's output (not a slice of a
        # concat buffer, not through the upsample), stride 1, bf16.  The sums wait in bn_sums for the very next layer
        # of the backward pass.
        if self.fuse_bn_bwd_sums and dt == torch.bfloat16:
            writers: Dict[int, List[int]] = {}
            for i, L in enumerate(P.layers):
                if not L.frozen and L.dx is not None:
                    writers.setdefault(id(L.dx.buf), []).append(i)
            for i, L in enumerate(P.layers):
                if i == 0 or L.frozen or L.dx is None:
                    continue
                Q, c = P.layers[i - 1], L.conv
                if (c.stride == 1 and c.in_channels > 32 and c.in_channels % 8 == 0 and not Q.frozen and Q.z is not None
                        and Q.up_dst is None and Q.y.buf is L.x.buf and Q.y.ch_offset == L.x.ch_offset
                        and Q.y.C == L.x.C == L.x.buf.shape[3] and min(writers[id(L.dx.buf)]) == i):
                    L.sums_for = Q
        # every parameter gradient is a view of one flat arena, in the order backward produces them, so that
        # data-parallel training can all-reduce finished ranges while backward continues (utils/dist.py); the arena
        # belongs to the engine (all plans share it) and the parameters themselves live in its twin
        order = [L for L in reversed(P.layers) if not L.frozen]
        if self.grad_arena is None:
            self._adopt(order)
        assert [L.name for L in order] == self._order
        P.arena = self.grad_arena
        for L in order:
            c = L.conv
            off, end = self._slots[L.name]
            n = c.weight.numel()
            L.dW = P.arena[off:off + n].view(c.out_channels, c.kernel_size, c.kernel_size, c.in_channels)
            off += n
            if c.norm is not None:
                L.dgamma = P.arena[off:off + c.out_channels]
                L.dbeta = P.arena[off + c.out_channels:off + 2 * c.out_channels]
            else:
                L.dbias = P.arena[off:off + c.out_channels]
            L.arena_end = end
        n_bn = max(ops.bn_workspace_bytes(N, L.y.H, L.y.W, L.conv.out_channels) for L in P.layers if L.z is not None)
        P.bn_ws = torch.empty(n_bn, dtype=torch.uint8, device=dev)
        n_sums = max([ops.conv_bnsums_rows(L.x, L.z, L.conv.in_channels, L.conv.out_channels, L.conv.kernel_size,
                                           L.conv.stride) * 2 * L.conv.out_channels
                      for L in P.layers if self._sums_in_epilogue(L)], default=0)
        n_sums = max([n_sums] + [ops.conv_bnsums_rows(L.dz, L.dx, (L.conv.out_channels + 63) // 64 * 64,
                                                      L.conv.in_channels, L.conv.kernel_size) * 2 * L.conv.in_channels
                                 for L in P.layers if L.sums_for is not None])
        P.bn_sums = torch.empty(max(n_sums, 1), device=dev)
        n_wg = max(ops.stem_wgrad_workspace_bytes() if L.stem else
                   ops.wgrad_workspace_bytes(N * L.y.H * L.y.W, L.conv.in_channels, L.conv.out_channels, L.conv.kernel_size)
                   for L in P.layers if not L.frozen)
        P.wg_ws = torch.empty(n_wg, dtype=torch.uint8, device=dev)
        return P

    def _sums_in_epilogue(self, L: TLayer) -> bool:
        """Whether the layer's BatchNorm sums come out of its convolution's epilogue.  The thin first-stage layers
        (32 output channels, or 32 input channels under a 3x3 bf16 filter: kernel instances without that epilogue) and
        the stem take the separate statistics pass."""
        c = L.conv
        return (self.fuse_bn_sums and L.z is not None and not L.stem and c.out_channels > 32
                and not (self.dtype == torch.bfloat16 and c.in_channels == 32 and c.kernel_size == 3))

    @staticmethod
    def _is_conv1(L: TLayer) -> bool:
        return L.name.endswith(".conv1") and ".res" in L.name

    def _producer_frozen(self, P: TPlan, L: TLayer) -> bool:
        for M in P.layers:
            if M.y.buf is L.x.buf and M is not L:
                return M.frozen
        return True  # produced by the fused frozen prefix

    # ------------------------------------------------------------------ forward
    @ops.on_own_device
    def forward(self, images: torch.Tensor, mean, std, *, update_running_stats: bool = True) -> List[View]:
        """images [N,3,Hs,Ws] uint8/float32 (0..255) on the GPU; returns the head views p3, p4, p5 (f32 NHWC)."""
        N, _, Hs, Ws = images.shape
        P = self.plan(N, (Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32)
        self._pack_all()
        images = images.contiguous()
        stem_trainable = bool(P.layers) and P.layers[0].stem
        if stem_trainable:
            P.images, P.norm = images, (tuple(mean), tuple(std))  # backward normalises the images again for dW
        else:
            stem_w, stem_b = self._packed("bottom_up.stem.conv", "folded")
            if P.stem_y is not None:
                dw, db = self._packed("bottom_up.res1.0.downsample", "folded")
                ops.stem_down_fwd(images, stem_w, stem_b, dw, db, P.stem_y, padded_hw=(P.H, P.W), mean=mean, std=std)
                for step in P.prefix:
                    step()
            else:
                ops.stem_fwd(images, stem_w, stem_b, P.stem_out, mean=mean, std=std)
        for L in P.layers:
            c = L.conv
            if L.stem:   # normalisation + 3 -> 32 convolution from the images, then batch statistics like any layer
                n = c.norm
                rm, rv = (n.running_mean, n.running_var) if update_running_stats else (None, None)
                ops.stem_fwd(images, self._stem_w, self._zeros(c.out_channels), L.z, mean=mean, std=std, slope=1.0)
                ops.bn_train_stats(L.z, n.weight.detach(), n.bias.detach(), L.stats, P.bn_ws, eps=n.eps,
                                   momentum=BN_MOMENTUM, running_mean=rm, running_var=rv)
                ops.bn_act_fwd(L.z, L.stats, L.y, slope=L.slope)
                continue
            if L.frozen:
                w, b = self._packed(L.name, "folded")
                ops.conv2d_fwd(L.x, w, b, L.y, cin=c.in_channels, cout=c.out_channels, k=c.kernel_size, stride=c.stride,
                               slope=L.slope, residual=L.residual)
                continue
            w, b = self._packed(L.name, "raw")
            if c.norm is None:   # linear head with bias
                ops.conv2d_fwd(L.x, w, b, L.y, cin=c.in_channels, cout=c.out_channels, k=c.kernel_size, stride=c.stride,
                               slope=1.0)
                continue
            n = c.norm
            rm, rv = (n.running_mean, n.running_var) if update_running_stats else (None, None)
            if self._sums_in_epilogue(L):   # the conv epilogue leaves per-tile sums of z and z*z: no second pass over z
                rows = ops.conv2d_fwd_bnsums(L.x, w, b, L.z, P.bn_sums, cin=c.in_channels, cout=c.out_channels,
                                             k=c.kernel_size, stride=c.stride)
                ops.bn_train_finalize(P.bn_sums, rows, N * L.z.H * L.z.W, n.weight.detach(), n.bias.detach(), L.stats,
                                      eps=n.eps, momentum=BN_MOMENTUM, running_mean=rm, running_var=rv)
            else:
                ops.conv2d_fwd(L.x, w, b, L.z, cin=c.in_channels, cout=c.out_channels, k=c.kernel_size, stride=c.stride,
                               slope=1.0)
                ops.bn_train_stats(L.z, n.weight.detach(), n.bias.detach(), L.stats, P.bn_ws, eps=n.eps,
                                   momentum=BN_MOMENTUM, running_mean=rm, running_var=rv)
            ops.bn_act_fwd(L.z, L.stats, L.y, slope=L.slope, residual=L.residual)
            if L.up_dst is not None:
                ops.upsample2x(L.y, L.up_dst)
        return P.heads

    # ------------------------------------------------------------------ backward
    @ops.on_own_device
    def backward(self, P: TPlan, dheads: Optional[Sequence[torch.Tensor]]) -> None:
        """dheads: gradients of the loss w.r.t. the head tensors p3, p4, p5 as [N, A*(5+C), G, G] (any strides), or
        None when P.dheads already hold them (written by ops.train_loss_bwd).  Fills L.dW / dgamma / dbeta / dbias
        of every trainable layer."""
        ak = (5 + self.num_classes) * self.num_anchors
        head_grad = {}
        if dheads is not None:
            for dv, g in zip(P.dheads, dheads):
                dv.interior()[..., :ak].copy_(g.permute(0, 2, 3, 1))
            head_grad = {id(dv.buf): g for dv, g in zip(P.dheads, dheads)}
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1
        if ddp:
            if P.bucketer is None:
                from .utils.dist import GradBucketer
                P.bucketer = GradBucketer(P.arena, self.bucket_bytes, self.process_group)
            P.bucketer.start()
        main = torch.cuda.current_stream(self.device)
        side = self._wgrad_stream() if self.wgrad_on_side_stream else main
        for L in reversed(P.layers):
            if L.frozen:
                continue
            c = L.conv
            if c.norm is None:
                g = head_grad.get(id(L.dz.buf))
                if g is not None:
                    L.dbias.copy_(g.sum(dim=(0, 2, 3)))
                else:
                    torch.sum(L.dz.interior()[..., :ak], dim=(0, 1, 2), dtype=torch.float32, out=L.dbias)
            else:
                if L.up_src is not None:
                    ops.sum2x2(L.up_src, L.dy)
                if L.presummed_rows:   # the launch that wrote dy (the previous step of this loop) left the reduction
                    ops.bn_act_bwd_presummed(L.dy, L.z, L.stats, L.dz, L.dgamma, L.dbeta, P.bn_sums, L.presummed_rows,
                                             slope=L.slope)
                    L.presummed_rows = 0
                else:
                    ops.bn_act_bwd(L.dy, L.z, L.stats, L.dz, L.dgamma, L.dbeta, P.bn_ws, slope=L.slope)
            dz = View(L.dz.buf, L.dz.H, L.dz.W, L.dz.halo, (c.out_channels + 7) // 8 * 8, 0)
            # The weight gradient only feeds the optimizer: it runs on a stream of its own, next to the data gradient
            # and the memory-bound BatchNorm backward of the following layers (it reads x and dz, which nothing
            # overwrites during backward, and shares its workspace only with the other weight gradients, which
            # the side stream keeps in order).
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                if L.stem:
                    ops.stem_wgrad(P.images, dz, L.dW, P.wg_ws, mean=P.norm[0], std=P.norm[1],
                                   round_input_bf16=self.dtype == torch.bfloat16)
                else:
                    ops.conv2d_wgrad(L.x, dz, L.dW, P.wg_ws, cin=c.in_channels, cout=c.out_channels, k=c.kernel_size,
                                     stride=c.stride)
                if ddp:
                    P.bucketer.ready(L.arena_end)  # this layer's gradients are final: maybe launch a bucket
            if L.dx is None:
                continue
            w, b = self._packed(L.name, "dgrad")
            cpad = (c.out_channels + 63) // 64 * 64
            src = View(L.dz.buf, L.dz.H, L.dz.W, L.dz.halo, cpad, 0)
            res = L.dx_add if L.dx_add is not None else (L.dx if L.dx_accumulate else None)
            if L.sums_for is not None:
                Q = L.sums_for
                Q.presummed_rows = ops.conv2d_fwd_bnbwd_sums(src, w, b, L.dx, P.bn_sums, cin=cpad, cout=c.in_channels,
                                                             k=c.kernel_size, z=Q.z, stats=Q.stats, act_slope=Q.slope,
                                                             residual=res)
                continue
            if c.stride == 2 and L.dzd is None:   # bf16: straight from dz, only the taps that reach each output pixel
                ops.conv3x3s2_dgrad(src, w, b, L.dx, cin=cpad, cout=c.in_channels, residual=res)
                continue
            if c.stride == 2:
                ops.dilate2x(src, L.dzd)
                src = L.dzd
            ops.conv2d_fwd(src, w, b, L.dx, cin=cpad, cout=c.in_channels, k=c.kernel_size, stride=1, slope=1.0, residual=res)
        if side is not main:
            main.wait_stream(side)
        if ddp:
            P.bucketer.finish()

    def _wgrad_stream(self):
        if self._wstream is None:
            self._wstream = torch.cuda.Stream(self.device)
        return self._wstream

    def param_grads(self, P: TPlan) -> Dict[str, torch.Tensor]:
        """name -> gradient in the parameter's own layout ([Cout, Cin, k, k] views of the f32 buffers)."""
        out = {}
        for L in P.layers:
            if L.frozen:
                continue
            out[L.name + ".weight"] = L.dW.permute(0, 3, 1, 2)
            if L.conv.norm is not None:
                out[L.name + ".norm.weight"] = L.dgamma
                out[L.name + ".norm.bias"] = L.dbeta
            else:
                out[L.name + ".bias"] = L.dbias
        return out
