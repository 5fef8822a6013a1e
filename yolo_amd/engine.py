"""Whole-network inference plan: Darknet-53 + FPN neck + head decode + NMS as a fixed sequence of
gfx950 kernel launches over preallocated pixel-major (NHWC, zero-halo) buffers.

What is fused away relative to the reference's eager graph (all citations: reference checkout):
  * BatchNorm (folded into weights/bias) and LeakyReLU: epilogue of every conv
    (yolo/modeling/backbone/darknet.py:58-59,115-123; fpn.py:34-35,68-69,97-98);
  * ``residual = x.clone(); ...; out += residual`` (darknet.py:118,125): the conv2 launch adds the block
    input in its epilogue and writes IN PLACE over it (each element is read and written by one lane);
  * ``nn.Upsample`` + ``torch.cat`` (fpn.py:36,179): the upsample conv writes each pixel 4x into channels
    [0, C/2) of the concat buffer, and the last block of res3/res4 writes its output straight into
    channels [C/2, ...) of the same buffer;
  * head view/slice/sigmoid/permute copies and per-image, per-level Python loops
    (yolo/modeling/meta_arch/yolo.py:266-383): one decode + one NMS call for the batch.
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import View
from .utils.plan_cache import PlanCache


@dataclass
class PackedConv:
    w: torch.Tensor
    b: torch.Tensor
    cin: int
    cout: int
    k: int
    stride: int


@dataclass
class Plan:
    N: int
    H: int
    W: int
    steps: List[Callable[[], None]] = field(default_factory=list)
    stem_out: Optional[View] = None   # None when the stem is fused into the first down-sampling conv
    stem_fused_y: Optional[View] = None  # output of the fused stem + res1.downsample launch
    pre_steps: List[Callable[[], None]] = field(default_factory=list)  # fused launches that run before `steps`
    heads: List[View] = field(default_factory=list)  # p3, p4, p5
    taps: Dict[str, View] = field(default_factory=dict)
    cand: Optional[ops.Candidates] = None
    decode_ws: Optional[torch.Tensor] = None
    nms_ws: Optional[torch.Tensor] = None
    splitk_ws: Dict[str, torch.Tensor] = field(default_factory=dict)  # partial-sum slabs of split-K layers, per stream
    keep: Optional[torch.Tensor] = None
    n_keep: Optional[torch.Tensor] = None
    cand_cap: int = 0  # capacity wanted for the candidate buffers (0: the default of 4096 per image)
    flops: int = 0  # conv multiply-add FLOPs (2 per MAC) of one forward at this batch
    meta: List[dict] = field(default_factory=list)  # per conv step: name, x, y, slope, residual, up


@dataclass
class InFlight:
    """A batch whose convolutions and decode are queued on the GPU (Yolov3Engine.submit)."""
    plan: Plan
    cand: ops.Candidates
    event: "torch.cuda.Event"
    score_thr: float
    geom: Tuple[int, int, int]  # N, stored image height, width
    # finish(record_count=True): the batch's candidate total, copied into pinned host memory behind the decode, and the
    # event that says the copy has landed (the throughput mode checks for overflow without making the host wait)
    count_host: Optional[torch.Tensor] = None
    count_event: Optional["torch.cuda.Event"] = None


@dataclass
class _GraphedDetect:
    """One captured detect() call: the HIP graph, its static input, its static outputs, the batch record of the
    captured call (plan and candidate buffers: kept alive with the graph) and the size tensors the gather reads."""
    graph: "torch.cuda.CUDAGraph"
    static_in: torch.Tensor
    det: torch.Tensor
    n_det: torch.Tensor
    in_flight: InFlight
    sizes: tuple
    buffers: tuple = ()   # keep / n_keep / NMS workspace the captured launches write: alive as long as the graph


class Yolov3Engine:
    """Packs a DarkNetFPN module's parameters once and runs batches through cached plans."""

    def __init__(self, backbone, *, num_classes: int, anchors_wh, strides: Sequence[int] = (8, 16, 32),
                 dtype: torch.dtype = torch.bfloat16, head_dtype: torch.dtype = torch.float32,
                 blocks: Sequence[int] = (1, 2, 8, 8, 4), fuse_stem: bool = True):
        self.backbone = backbone
        self.num_classes = num_classes
        self.num_anchors = len(anchors_wh[0])
        self.anchors_wh = anchors_wh
        self.strides = list(strides)
        self.dtype = dtype
        self.head_dtype = head_dtype
        self.blocks = list(blocks)
        self.fuse_stem = fuse_stem
        self.device = next(backbone.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("Yolov3Engine needs the model on an MI355X (cuda device); there is no CPU path")
        self._packed: Dict[str, PackedConv] = {}
        self._plans = PlanCache(self.device)  # (N, H, W, slot) -> Plan, least recently used first out
        self._in_hw: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._side = None
        self._dstream = None
        self._bstream = None
        self.branch_on_side_stream = os.environ.get("YOLO_BRANCH_STREAM", "1") != "0"
        self.nms_on_side_stream = os.environ.get("YOLO_NMS_STREAM", "1") != "0"
        self.decode_on_side_stream = os.environ.get("YOLO_DECODE_STREAM", "1") != "0"
        self.split_k = True  # few-pixel layers (small batches) split every tile's k loop over several workgroups
        # measurement hook: when a list, submit()/finish() append (label, start event, end event) recorded on the
        # stream the decode / NMS + gather launches run on
        self.trace_events: Optional[list] = None
        # detect_graphed(): call signature -> captured HIP graph of one detect() with its static input and outputs
        self._graphs: Dict[tuple, "_GraphedDetect"] = {}
        self._graph_seen: Dict[tuple, int] = {}
        self.last_in_flight: Optional[InFlight] = None
        self._count_host: Dict[int, torch.Tensor] = {}   # id(plan) -> pinned word for the candidate total
        self._lanes = None
        self.repack()

    # ------------------------------------------------------------------ weights
    @ops.on_own_device
    def repack(self) -> None:
        """(Re)build folded + packed weights from the module's current parameters."""
        self._packed.clear()
        for name, conv in self.backbone.darknet_modules():
            w, b = conv.folded()
            if conv.in_channels == 3:
                pw, pb = ops.pack_stem(w, b, self.device)
            else:
                pw, pb = ops.pack_conv(w, b, self.dtype, self.device)
            self._packed[name] = PackedConv(pw, pb, conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride)
        self._plans.clear()
        self._graphs.clear()   # captured launches point at the old packed weights

    # ------------------------------------------------------------------ plan
    @ops.on_own_device
    def plan(self, N: int, H: int, W: int, slot: int = 0) -> Plan:
        """Cached plan for a batch shape; plans with different ``slot`` own independent buffers."""
        return self._plans.get((N, H, W, slot), lambda: self._build_plan(N, H, W))

    def _build_plan(self, N: int, H: int, W: int) -> Plan:
        assert H % 32 == 0 and W % 32 == 0, "input must be padded to the backbone's size_divisibility (32)"
        dt, dev = self.dtype, self.device
        P = Plan(N, H, W)
        pk = self._packed

        def conv(name: str, x: View, y: View, slope: float = 0.1, residual: Optional[View] = None,
                 up: bool = False):
            c = pk[name]
            P.flops += 2 * N * (y.H // (2 if up else 1)) * (y.W // (2 if up else 1)) * c.cout * c.cin * c.k * c.k
            P.meta.append(dict(name=name, x=x, y=y, slope=slope, residual=residual, up=up))
            M = N * (y.H // (2 if up else 1)) * (y.W // (2 if up else 1))
            splits, nbytes = (ops.conv_splitk_plan(M, c.cin, c.cout, c.k, c.stride, dt)
                              if dt == torch.bfloat16 and self.split_k else (1, 0))
            if splits > 1:  # few-pixel layer (small batch): several workgroups share every tile's k loop
                # the output branches of the two coarse levels run on a stream of their own (run_backbone): their
                # partial sums must not share a buffer with the main stream's layers
                key = "branch" if name.startswith(("fpn_output0.", "fpn_output1.")) else "main"
                if key not in P.splitk_ws or P.splitk_ws[key].numel() < nbytes:
                    P.splitk_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                P.steps.append(lambda: ops.conv2d_fwd_splitk(x, c.w, c.b, y, P.splitk_ws[key], cin=c.cin, cout=c.cout,
                                                             k=c.k, stride=c.stride, slope=slope, residual=residual,
                                                             upsample2x=up))
                return
            P.steps.append(lambda: ops.conv2d_fwd(x, c.w, c.b, y, cin=c.cin, cout=c.cout, k=c.k, stride=c.stride,
                                                  slope=slope, residual=residual, upsample2x=up))

        stem = pk["bottom_up.stem.conv"]
        down0 = pk["bottom_up.res1.0.downsample"]
        # bf16: stem and the first stride-2 conv run as one launch, the stem output never reaches HBM
        fused = (self.fuse_stem and dt == torch.bfloat16 and stem.cout == 32 and down0.cout == 64
                 and down0.k == 3 and down0.stride == 2)
        if not fused:
            P.stem_out = ops.alloc_view(N, H, W, stem.cout, 1, dt, dev)
        P.flops += 2 * N * H * W * stem.cout * 27
        x = P.stem_out
        h, w, c_in = H, W, stem.cout
        c_out = pk["bottom_up.res1.0.downsample"].cout
        stage_out: Dict[int, View] = {}
        # channels of res3/res4/res5 decide the concat buffers of the neck
        chans = [c_out * 2 ** i for i in range(5)]
        cat = {
            3: ops.alloc_view(N, H // 8, W // 8, chans[2] + chans[2] // 2, 1, dt, dev),   # [up(128) | res3(256)]
            4: ops.alloc_view(N, H // 16, W // 16, chans[3] + chans[3] // 2, 1, dt, dev),  # [up(256) | res4(512)]
        }
        for si, nb in enumerate(self.blocks, start=1):
            h, w = h // 2, w // 2
            co = chans[si - 1]
            X = ops.alloc_view(N, h, w, co, 1, dt, dev)
            T = ops.alloc_view(N, h, w, co // 2, 1, dt, dev)
            if si == 1 and fused:
                P.stem_fused_y = X
                P.flops += 2 * N * h * w * down0.cout * down0.cin * 9
            else:
                conv(f"bottom_up.res{si}.0.downsample", x, X)
            # bf16, 64 channels: each residual block is one launch (the 32-channel intermediate stays in LDS),
            # ping-ponging between two buffers because that launch cannot run in place
            fuse_block = (self.fuse_stem and fused and si == 1 and co == 64 and si not in cat)
            X2 = ops.alloc_view(N, h, w, co, 1, dt, dev) if fuse_block else None
            for b in range(nb):
                p = f"bottom_up.res{si}.{b}"
                if fuse_block:
                    c1, c2 = pk[p + ".conv1"], pk[p + ".conv2"]
                    P.flops += 2 * N * h * w * (c1.cout * c1.cin + c2.cout * c2.cin * 9)
                    P.pre_steps.append(lambda c1=c1, c2=c2, a=X, o=X2: ops.resblock_c64_fwd(a, c1.w, c1.b, c2.w, c2.b, o))
                    X, X2 = X2, X
                    continue
                conv(p + ".conv1", X, T)
                last = b == nb - 1
                if last and si in cat:
                    dst = cat[si].slice(co // 2, co)
                    conv(p + ".conv2", T, dst, residual=X)
                    X = dst
                else:
                    conv(p + ".conv2", T, X, residual=X)  # in place: y == residual
            stage_out[si] = X
            x = X
            P.taps[f"res{si}"] = X
        # ---- neck + heads, coarse to fine (fpn.py:165-185)
        ak = (5 + self.num_classes) * self.num_anchors
        ak_pad = (ak + 7) // 8 * 8
        heads = []
        feats = [stage_out[5], cat[4], cat[3]]
        for idx in range(3):
            f = feats[idx]
            lat = pk[f"fpn_lateral{idx}.blocks.0"].cout
            fh, fw = f.H, f.W
            A = ops.alloc_view(N, fh, fw, lat, 1, dt, dev)
            B = ops.alloc_view(N, fh, fw, lat * 2, 1, dt, dev)
            src = f
            for i in range(5):
                dst = A if i % 2 == 0 else B
                conv(f"fpn_lateral{idx}.blocks.{i}", src, dst)
                src = dst
            conv(f"fpn_output{idx}.conv", A, B)
            hv = ops.alloc_view(N, fh, fw, ak_pad, 0, self.head_dtype, dev)
            conv(f"fpn_output{idx}.head", B, hv, slope=1.0)
            heads.insert(0, View(hv.buf, fh, fw, 0, ak, 0))
            if idx < 2:
                up_c = pk[f"fpn_upsample{idx}.conv"].cout
                conv(f"fpn_upsample{idx}.conv", A, cat[4 - idx].slice(0, up_c), up=True)
        P.heads = heads
        P.taps.update(p3=heads[0], p4=heads[1], p5=heads[2])
        return P

    # ------------------------------------------------------------------ execution
    @ops.on_own_device
    def run_backbone(self, images: torch.Tensor, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0),
                     padded_hw: Optional[Tuple[int, int]] = None, plan: Optional[Plan] = None) -> Plan:
        """images: [N,3,Hs,Ws] float32 or uint8 on the GPU.  Pixel normalisation with (mean, std) and
        zero padding up to ``padded_hw`` happen inside the stem kernel."""
        N, _, Hs, Ws = images.shape
        if padded_hw is None:
            padded_hw = ((Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32)
        P = plan if plan is not None else self.plan(N, padded_hw[0], padded_hw[1])
        self.run_backbone_prefix(images, mean, std, P)
        if self.trace_events is not None:   # measurement: an event pair around every convolution launch, where it runs
            steps = [lambda i=i, step=step: self._traced(("conv", i), step) for i, step in enumerate(P.steps)]
        else:
            steps = P.steps
        if not self.branch_on_side_stream:
            for step in steps:
                step()
            return P
        # The output branches of the two coarse levels (3x3 conv + head conv, fpn.py:96-100) depend only on their
        # level's lateral block and write buffers nothing else touches: they run on a stream of their own next to
        # the upsample conv and the next level's lateral block (few-tile layers that leave CUs idle on their own).
        main = torch.cuda.current_stream(self.device)
        side = self._branch_stream()
        for meta, step in zip(P.meta, steps):
            name = meta["name"]
            if name.startswith(("fpn_output0.", "fpn_output1.")):
                if name.endswith(".conv"):
                    side.wait_stream(main)
                with torch.cuda.stream(side):
                    step()
            else:
                step()
        main.wait_stream(side)
        return P

    @ops.on_own_device
    def run_backbone_prefix(self, images: torch.Tensor, mean, std, P: Plan) -> None:
        """The launches that precede ``P.steps``: stem (fused with the first down-sampling conv in bf16) and the
        fused residual blocks of the first stage."""
        stem = self._packed["bottom_up.stem.conv"]
        if P.stem_fused_y is not None:
            d = self._packed["bottom_up.res1.0.downsample"]
            ops.stem_down_fwd(images.contiguous(), stem.w, stem.b, d.w, d.b, P.stem_fused_y,
                              padded_hw=(P.H, P.W), mean=mean, std=std)
        else:
            ops.stem_fwd(images.contiguous(), stem.w, stem.b, P.stem_out, mean=mean, std=std)
        for step in P.pre_steps:
            step()

    @ops.on_own_device
    def decode(self, P: Plan, score_thr: float, cap: Optional[int] = None) -> ops.Candidates:
        if cap is None:
            cap = max(P.cand_cap, P.N * 4096)
        if P.cand is None or P.cand.scores.numel() < cap:
            dev = self.device
            self._drop_graphs_of(P)   # captured launches of this plan write the old candidate buffers
            P.cand = ops.Candidates(torch.empty((cap, 4), dtype=torch.float32, device=dev),
                                    torch.empty((cap,), dtype=torch.float32, device=dev),
                                    torch.empty((cap,), dtype=torch.int32, device=dev),
                                    torch.empty((P.N + 1,), dtype=torch.int32, device=dev))
            P.decode_ws = torch.empty(ops.decode_workspace_bytes(P.heads), dtype=torch.uint8, device=dev)
        return ops.decode(P.heads, self.strides, self.anchors_wh, self.num_classes, score_thr,
                          P.cand.scores.numel(), out=P.cand, workspace=P.decode_ws)

    @ops.on_own_device
    def submit(self, images: torch.Tensor, *, score_thr: float, mean, std, padded_hw=None, slot: int = 0) -> "InFlight":
        """First half of detect(): enqueue backbone + neck + heads + decode for a batch.  Nothing here waits for the
        GPU.  ``slot`` selects one of several independent sets of activation buffers, so that batch i+1 can be
        submitted before batch i is finished (its post-processing then runs next to batch i+1's convolutions)."""
        N, _, Hs, Ws = images.shape
        if padded_hw is None:
            padded_hw = ((Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32)
        P = self.plan(N, padded_hw[0], padded_hw[1], slot)
        self.run_backbone(images, mean, std, padded_hw, plan=P)
        # The decode launches (small, latency-bound) go to their own stream behind the batch's last convolution, so
        # the next batch's convolutions follow this batch's directly on the main stream.  The head tensors of this
        # slot are not written again before the main stream has waited for this slot's post-processing (finish()).
        main = torch.cuda.current_stream(self.device)
        dstream = self._decode_stream() if (self.nms_on_side_stream and self.decode_on_side_stream) else main
        if dstream is not main:
            dstream.wait_stream(main)
        with torch.cuda.stream(dstream):
            t0 = self._trace_begin()
            cand = self.decode(P, score_thr)
            self._trace_end("decode", t0)
            ev = torch.cuda.Event()
            ev.record()
        return InFlight(P, cand, ev, score_thr, (N, Hs, Ws))

    @ops.on_own_device
    def finish(self, h: "InFlight", *, nms_thr: float, max_det: int = 100, nms_mode: int = ops.NMS_AUTO,
               auto_limit: int = 20000, in_hw: Optional[torch.Tensor] = None, out_hw: Optional[torch.Tensor] = None,
               record_count: bool = False):
        """Second half of detect(): enqueue batched NMS + top-k gather behind the batch's decode.  The candidate
        counts stay on the device (the NMS reads them there), so nothing is read back and the host never waits.
        Returns (det [N, max_det, 6], n_det [N]).  A batch with more candidates than the candidate buffers hold is
        processed on the first ``capacity`` of them; ``overflowed(h)`` tells (it reads one word back), and with
        ``record_count`` the total is also copied to pinned host memory behind the decode (``h.count_host`` is valid
        once ``h.count_event`` has completed) so that a caller can check later without waiting now."""
        P, cand = h.plan, h.cand
        cap = cand.scores.numel()
        dev = self.device
        # NMS + gather go to a side stream: a few launch-bound kernels that would otherwise sit between two batches'
        # convolutions on the main stream.  The main stream waits for them before it goes on, which orders both the
        # caller's use of the result and the next reuse of this slot's buffers -- at that point the main stream is
        # still busy with the convolutions queued by the following submit(), so the wait costs nothing.
        main = torch.cuda.current_stream(dev)
        side = self._side_stream() if self.nms_on_side_stream else main
        if side is not main:
            side.wait_event(h.event)
        with torch.cuda.stream(side):
            t0 = self._trace_begin()
            if record_count:
                word = self._count_host.get(id(P))
                if word is None:
                    if len(self._count_host) > 64:
                        self._count_host.clear()
                    word = self._count_host[id(P)] = torch.zeros(1, dtype=torch.int32).pin_memory()
                word.copy_(cand.img_offsets[P.N:P.N + 1], non_blocking=True)
                h.count_host = word
                h.count_event = torch.cuda.Event()
                h.count_event.record()
            if P.keep is None or P.keep.shape[1] != max_det:
                self._drop_graphs_of(P)   # captured launches of this plan write the old keep buffers
                P.keep = torch.empty((P.N, max_det), dtype=torch.int32, device=dev)
                P.n_keep = torch.empty((P.N,), dtype=torch.int32, device=dev)
            need = ops.nms_workspace_bytes(P.N, cap)
            if P.nms_ws is None or P.nms_ws.numel() < need:
                self._drop_graphs_of(P)
                P.nms_ws = torch.empty(need, dtype=torch.uint8, device=dev)
            ops.batched_nms(cand.boxes, cand.scores, cand.classes, cand.img_offsets, cap, nms_thr,
                            num_classes=self.num_classes, mode=nms_mode, auto_limit=auto_limit, top_k=max_det,
                            keep_stride=max_det, workspace=P.nms_ws, keep=P.keep, n_keep=P.n_keep)
            N, Hs, Ws = h.geom
            if in_hw is None:  # same stored size for the whole batch: cached, no host-to-device copy per call
                if h.geom not in self._in_hw:
                    self._in_hw[h.geom] = torch.tensor([[Hs, Ws]] * N, dtype=torch.int32, device=dev)
                in_hw = self._in_hw[h.geom]
            if out_hw is None:
                out_hw = in_hw
            det, n_det = ops.gather_detections(cand, P.keep, P.n_keep, in_hw, out_hw)
            self._trace_end("nms_gather", t0)
        if side is not main:
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():   # (a captured graph owns its outputs)
                det.record_stream(main)
                n_det.record_stream(main)
        return det, n_det

    def grow_candidates(self, h: "InFlight", total: Optional[int] = None) -> bool:
        """After a batch: if it produced more candidates than its buffers hold, enlarge the plan's buffers (for the
        next run) and return True -- the caller runs the batch again.  ``total``: the batch's candidate count
        (``img_offsets[N]``) if the caller has read it already; otherwise it is read here, which waits for the decode."""
        over = self.overflowed(h) if total is None else max(0, int(total) - h.cand.scores.numel())
        if over:
            h.plan.cand_cap = int((h.cand.scores.numel() + over) * 1.25) + 1024
            self._drop_graphs_of(h.plan)   # captured graphs of this plan hold the old candidate buffers
        return over > 0

    def _drop_graphs_of(self, P: Plan) -> None:
        """Forget the captured detect() calls that run on plan ``P``: their launches hold raw pointers of its buffers."""
        if torch.cuda.is_current_stream_capturing():
            return   # the capture in progress is not in the table yet; it was preceded by an eager call on this plan
        for key in [k for k, g in self._graphs.items() if g.in_flight.plan is P]:
            del self._graphs[key]

    def overflowed(self, h: "InFlight") -> int:
        """Candidates the batch produced beyond the capacity of its candidate buffers (0 = none were dropped).
        Reads one word back from the device, i.e. waits for the batch's decode."""
        total = int(h.cand.img_offsets[h.plan.N].item())
        return max(0, total - h.cand.scores.numel())

    def _traced(self, label, fn) -> None:
        t0 = self._trace_begin()
        fn()
        self._trace_end(label, t0)

    def _trace_begin(self):
        if self.trace_events is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _trace_end(self, label, start) -> None:
        if start is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace_events.append((label, start, ev))

    def _decode_stream(self):
        if self._dstream is None:
            self._dstream = torch.cuda.Stream(self.device)
        return self._dstream

    def _branch_stream(self):
        if self._bstream is None:
            self._bstream = torch.cuda.Stream(self.device)
        return self._bstream

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def detect(self, images: torch.Tensor, *, score_thr: float, nms_thr: float, mean, std,
               padded_hw=None, max_det: int = 100, nms_mode: int = ops.NMS_AUTO, auto_limit: int = 20000,
               in_hw: Optional[torch.Tensor] = None, out_hw: Optional[torch.Tensor] = None):
        """Full inference for a batch.  Returns (det [N, max_det, 6], n_det [N]) on the GPU:
        x0,y0,x1,y1 (output frame), score, class."""
        h = self.submit(images, score_thr=score_thr, mean=mean, std=std, padded_hw=padded_hw)
        out = self.finish(h, nms_thr=nms_thr, max_det=max_det, nms_mode=nms_mode, auto_limit=auto_limit,
                          in_hw=in_hw, out_hw=out_hw)
        self.last_in_flight = h
        return out

    GRAPH_AFTER = 3    # detect_graphed() captures a call signature when it comes for the third time
    GRAPH_KEEP = 8     # captured signatures kept (least recently used first out)

    @ops.on_own_device
    def detect_graphed(self, images: torch.Tensor, *, score_thr: float, nms_thr: float, mean, std,
                       padded_hw=None, max_det: int = 100, nms_mode: int = ops.NMS_AUTO, auto_limit: int = 20000,
                       in_hw: Optional[torch.Tensor] = None, out_hw: Optional[torch.Tensor] = None):
        """detect() for latency-bound callers (one or a few images per call, the same shapes again and again): the
        ~130 launches of a call -- convolutions, split-k sums, decode, NMS, gather, with their stream forks and joins --
        are captured ONCE per call signature into a HIP graph and replayed with a single launch, so the host's share of
        a call no longer grows with the number of layers.  A signature is captured when it has been seen
        ``GRAPH_AFTER`` times (calls before that, and signatures that keep changing, run detect() as it is); the input
        is copied into the graph's static input buffer, the returned tensors are the graph's static outputs: valid
        until the next call with the same signature, like the buffers of a plan.  ``last_in_flight`` is set as by
        detect()."""
        kw = dict(score_thr=score_thr, nms_thr=nms_thr, mean=mean, std=std, padded_hw=padded_hw, max_det=max_det,
                  nms_mode=nms_mode, auto_limit=auto_limit, in_hw=in_hw, out_hw=out_hw)
        key = (tuple(images.shape), images.dtype, float(score_thr), float(nms_thr), tuple(float(m) for m in mean),
               tuple(float(v) for v in std), padded_hw, int(max_det), int(nms_mode), int(auto_limit),
               None if in_hw is None else in_hw.data_ptr(), None if out_hw is None else out_hw.data_ptr())
        g = self._graphs.pop(key, None)
        if g is None:
            seen = self._graph_seen.get(key, 0) + 1
            if len(self._graph_seen) > 256:
                self._graph_seen.clear()
            self._graph_seen[key] = seen
            if seen < self.GRAPH_AFTER:
                return self.detect(images, **kw)
            static_in = images.clone()
            # plan, candidate buffers, workspaces, side streams and kernel attributes exist after an eager call; the
            # capture itself must allocate nothing but its outputs
            self.detect(static_in, **kw)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                det, n_det = self.detect(static_in, **kw)
            P = self.last_in_flight.plan
            g = _GraphedDetect(graph, static_in, det, n_det, self.last_in_flight, (in_hw, out_hw),
                               (P.keep, P.n_keep, P.nms_ws, P.decode_ws))
            while len(self._graphs) >= self.GRAPH_KEEP:
                self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g   # most recently used last
        g.static_in.copy_(images, non_blocking=True)
        g.graph.replay()
        self.last_in_flight = g.in_flight
        return g.det, g.n_det

    def detect_pipelined(self, batches, *, score_thr: float, nms_thr: float, mean, std, max_det: int = 100,
                         nms_mode: int = ops.NMS_AUTO, auto_limit: int = 20000, depth: int = 2,
                         lanes: Optional[int] = None):
        """Throughput mode over an iterable of image batches: batch i+1's convolutions are queued before batch i's
        NMS, on alternating sets of buffers, and the host never waits for the GPU.  Yields (det, n_det) per batch, in
        order.

        Candidate overflow: a batch's candidate total is copied to pinned host memory behind its decode and looked at
        one batch later (when the copy has long landed), BEFORE the batch's result is yielded -- results therefore leave
        one batch later than they are finished.  A batch that produced more candidates than its buffers hold (its NMS
        then saw only the first ``capacity`` of them: the last images of the batch lose detections) is run again on
        enlarged buffers of a plan of its own, and the enlarged capacity is kept for the batches that follow.

        ``lanes`` (default: YOLO_LANES or 1): with 2, consecutive batches alternate between two main streams, so two
        batches' convolutions are on the GPU at once and each launch's partly filled last round of workgroups is
        filled by the other batch's launch (measured + 1.9 % images/s at batch 64; every kernel then shares the chip,
        so per-kernel durations no longer say what a kernel does alone -- measurements of single kernels use one lane)."""
        if lanes is None:
            lanes = int(os.environ.get("YOLO_LANES", "1"))
        n_lanes = max(1, min(int(lanes), 2))
        depth = max(depth, n_lanes)
        pending: List[Tuple[InFlight, torch.Tensor]] = []            # submitted, NMS not queued yet
        done: List[Tuple[InFlight, torch.Tensor, tuple]] = []        # NMS queued, candidate total not looked at yet
        kw = dict(nms_thr=nms_thr, max_det=max_det, nms_mode=nms_mode, auto_limit=auto_limit)
        lane_done: Dict[int, "torch.cuda.Event"] = {}

        def settle(rec):
            h, images, out = rec
            if id(h) in lane_done:         # two lanes: the caller's stream has not seen this batch's lane yet
                torch.cuda.current_stream(self.device).wait_event(lane_done.pop(id(h)))
            h.count_event.synchronize()   # recorded a whole batch ago
            total = int(h.count_host.item())
            if total <= h.cand.scores.numel():
                return out
            # rare: run the batch again, alone, on a plan nothing in flight uses, with room for what it produced
            redo = self.plan(h.plan.N, h.plan.H, h.plan.W, depth)
            redo.cand_cap = max(redo.cand_cap, int(total * 1.25) + 1024)
            for s in range(depth):         # later batches of this shape start with the larger buffers too
                P = self.plan(h.plan.N, h.plan.H, h.plan.W, s)
                P.cand_cap = max(P.cand_cap, redo.cand_cap)
            h2 = self.submit(images, score_thr=score_thr, mean=mean, std=std, slot=depth)
            return self.finish(h2, **kw)

        lanes = None
        if n_lanes == 2 and self.trace_events is None:
            if self._lanes is None:
                self._lanes = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)]
            lanes = self._lanes
            caller = torch.cuda.current_stream(self.device)
            for l in lanes:
                l.wait_stream(caller)
        lane_of = {}

        def on_lane(i, fn, h=None):
            if lanes is None:
                return fn()
            with torch.cuda.stream(lanes[i % 2]):
                out = fn()
                if h is not None:          # the batch is complete on its lane behind this event
                    ev = torch.cuda.Event()
                    ev.record()
                    lane_done[id(h)] = ev
                return out

        for i, images in enumerate(batches):
            h = on_lane(i, lambda: self.submit(images, score_thr=score_thr, mean=mean, std=std, slot=i % depth))
            lane_of[id(h)] = i
            pending.append((h, images))
            if len(pending) == depth:
                h, im = pending.pop(0)
                done.append((h, im, on_lane(lane_of.pop(id(h)), lambda: self.finish(h, record_count=True, **kw), h)))
            if len(done) == 2:
                yield settle(done.pop(0))
        while pending:
            h, im = pending.pop(0)
            done.append((h, im, on_lane(lane_of.pop(id(h)), lambda: self.finish(h, record_count=True, **kw), h)))
        if lanes is not None:
            for l in lanes:
                torch.cuda.current_stream(self.device).wait_stream(l)
        while done:
            yield settle(done.pop(0))
