# Rehearsal of the 3-rank training loop on a one-GPU box: the ranks share the card and exchange gradients through
# gloo on host copies (production: one rank per GPU, RCCL).  usage: bash tools/rehearse_ddp.sh <direct|allreduce>
mode=${1:-direct}
cd "$GRAFT_REPO_ROOT" || exit 1
export YOLO_DIST_BACKEND=gloo YOLO_GRAD_EXCHANGE=$mode HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29611 \
    tools/train_bench.py --batch 4 --size 256 --steps 3 --warmup 1
