python bench.py --force-sharded --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | cut -c1-120
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | cut -c1-120
