"""RCCL rehearsal on whatever GPUs the box has (one rank per GPU; one GPU = a world of one): the collectives bench.py uses at N > 1 --
barrier, all_reduce MAX / SUM on device tensors, the end gather of word sequences (voicebridge_amd/dist.py) -- over backend "nccl".
usage (under gpurun): python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/rccl_smoke.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from voicebridge_amd import dist as vdist  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
tt = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
fr = torch.tensor([1000 * (rank + 1)], dtype=torch.int64, device=dev)
dist.all_reduce(fr, op=dist.ReduceOp.SUM)
every = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(world)]
dist.all_gather(every, torch.tensor([rank, 2.0, 3.0], dtype=torch.float64, device=dev))
ids = [10 * rank + i for i in range(3)]
hyps = [np.arange(i + 1, dtype=np.int32) + rank for i in range(3)]
totals, merged = vdist.gather_results(ids, hyps, np.arange(7, dtype=np.int64), device=dev)
torch.cuda.synchronize()
if rank == 0:
    ok = (float(tt.item()) == 1.5 + world - 1 and int(fr.item()) == 1000 * world * (world + 1) // 2 and len(merged) == 3 * world
          and list(totals) == [int(x) * world for x in range(7)] and [int(e[0].item()) for e in every] == list(range(world)))
    print(json.dumps({"backend": dist.get_backend(), "world_size": world, "devices": torch.cuda.device_count(), "ok": bool(ok)}))
dist.barrier()
dist.destroy_process_group()
sys.exit(0)
