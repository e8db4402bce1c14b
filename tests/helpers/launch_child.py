"""Child of tests/test_launch.py: one rank started by raytracer_amd.launch.spawn_ranks.  Joins a gloo group from the
environment the launcher set, all-reduces its rank, writes what it saw, and exits with the status asked for."""
import json
import os
import sys

import torch
import torch.distributed as dist

out_dir, fail_rank, fail_code = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == fail_rank and fail_code and len(sys.argv) > 4 and sys.argv[4] == "early":
    sys.exit(fail_code)                       # dies before the rendezvous: the others must not hang forever
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t)
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
    json.dump({"rank": rank, "world": dist.get_world_size(), "sum": t.item(), "local_rank": int(os.environ["LOCAL_RANK"]),
               "master": os.environ["MASTER_ADDR"] + ":" + os.environ["MASTER_PORT"]}, f)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"ok": True, "world": world}), flush=True)
sys.exit(fail_code if rank == fail_rank else 0)
