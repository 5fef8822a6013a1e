"""Count the torch operators one training step issues (host-side view): finds per-step copies / clones that a
fused path should not need."""
import sys
import torch
sys.path.insert(0, ".")
import workload
from yolo_amd.trainer import Trainer, synthetic_batches


def main():
    model, cfg = workload.build_model(config="yolov3.yaml", dtype="bf16",
                                      opts=["OUTPUT_DIR", "", "SOLVER.CHECKPOINT_PERIOD", 0, "SOLVER.BASE_LR", 1e-4])
    data = next(synthetic_batches(cfg, 64, 416, seed=7, device=model.device))
    tr = Trainer(cfg, model, iter(lambda: data, None), log=lambda m: None)
    for _ in range(3):
        tr.run_step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        for _ in range(2):
            tr.run_step()
        torch.cuda.synchronize()
    for e in sorted(prof.key_averages(group_by_stack_n=6), key=lambda e: -e.count):
        if e.key in ("aten::item", "aten::lift_fresh") and e.count >= 4:
            print(e.count / 2, e.key, [fr for fr in e.stack if "yolo" in fr or "torch/optim" in fr][:5])
    rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count)
    for e in rows[:45]:
        print(f"{e.count/2:8.1f}/step  {e.key[:40]:40s} {str(e.input_shapes)[:90]}")


if __name__ == "__main__":
    main()
