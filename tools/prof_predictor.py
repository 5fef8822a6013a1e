import cProfile, pstats, sys, os, io
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tools")
import workload
from inference import Predictor
model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3.yaml", dtype="bf16", head_dtype="f32")
pred = Predictor.__new__(Predictor)
pred.cfg, pred.model = cfg, model.eval()
pred.min_size, pred.max_size, pred.input_format = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.FORMAT
img = np.random.default_rng(0).integers(0, 256, (416, 416, 3), dtype=np.uint8)
for _ in range(20): pred(img)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): pred(img)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
