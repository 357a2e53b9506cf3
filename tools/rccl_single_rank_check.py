#!/usr/bin/env python3
"""GPU-box check: the collective object of a sharded proof (plonk/native.py TorchExchange) over torch.distributed's "nccl" backend = RCCL, with the ranks this box allows.
One MI355X gives ONE rank (RCCL refuses two ranks on one device): what this run proves is that RCCL initialises on this image and moves the library's exchange buffers
(uint8 device tensors, all_gather_into_tensor, called from a non-main host thread like bench.py's extras) — not scaling.  usage: rccl_single_rank_check.py"""
import os, sys, threading, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from zk_dcap_verifier_amd.plonk.native import TorchExchange
x = TorchExchange(world, 1 << 20, "cuda")
x.send[:4096] = (torch.arange(4096, device="cuda") % 251 + rank).to(torch.uint8)
err = []
def run():
    try:
        torch.cuda.set_device(0)
        for _ in range(3):
            x.all_gather(x.send.data_ptr(), x.recv.data_ptr(), 4096)
    except BaseException as e:
        err.append(repr(e))
th = threading.Thread(target=run, daemon=True)
th.start(); th.join(60)
ok = (not th.is_alive()) and not err and all(bool((x.recv[r * 4096:(r + 1) * 4096] == (torch.arange(4096, device="cuda") % 251 + r).to(torch.uint8)).all()) for r in range(world))
s = torch.ones(8, device="cuda", dtype=torch.float64) * (rank + 1)
dist.all_reduce(s, op=dist.ReduceOp.MAX)
dist.barrier()
if rank == 0:
    print(json.dumps({"backend": "nccl (RCCL)", "ranks": world, "all_gather_into_tensor_uint8_from_a_helper_thread": ok, "errors": err, "all_reduce_max": float(s[0]), "calls": x.calls,
                      "seconds": round(time.time() - t0, 2), "torch": torch.__version__, "nccl_version": ".".join(map(str, torch.cuda.nccl.version()))}))
dist.destroy_process_group()
sys.exit(0 if ok else 1)
