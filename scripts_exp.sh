#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== 2 ranks on one GPU through torch.distributed.run (gloo rendezvous, library loaded before torch)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --reads 8000000 2>&1 | tail -3
echo "== 1 rank through torch.distributed.run"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -2
