import sys, time, torch
sys.path.insert(0, '.')
import workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16")
eng = model.backbone.engine()
imgs = workload.synthetic_images(N, 416, 416, seed=1, device="cuda")
mean, std = cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD
def step():
    return eng.detect(imgs, score_thr=0.25, nms_thr=0.5, mean=mean, std=std)
for _ in range(3): det, nd = step()
torch.cuda.synchronize()
P = eng.plan(N, 416, 416)
print("cands", int(P.cand.img_offsets[N]), "dets", nd.tolist()[:8], "flops/img", P.flops / N / 1e9)
t = time.time(); K = 10
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.time() - t) / K
print(f"N={N} {dt*1e3:.2f} ms/step  {N/dt:.0f} img/s  conv {P.flops/dt/1e12:.1f} TFLOP/s eff")
# backbone only
t = time.time()
for _ in range(K): eng.run_backbone(imgs, mean, std)
torch.cuda.synchronize()
dt = (time.time() - t) / K
print(f"backbone only {dt*1e3:.2f} ms  {P.flops/dt/1e12:.1f} TFLOP/s")
t = time.time()
for _ in range(K):
    Pp = eng.run_backbone(imgs, mean, std)
    eng.decode(Pp, 0.25)
torch.cuda.synchronize()
dt = (time.time() - t) / K
print(f"backbone+decode (no sync) {dt*1e3:.2f} ms")
