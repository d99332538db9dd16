"""Exercises pylamp_amd.parallel.Comm's start-up self-test of the NCCL (RCCL) transport on ONE rank (the only multi-process GPU
configuration this build environment has): python tools/nccl_selftest.py"""
import os, sys, datetime
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
import numpy as np
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
from pylamp_amd import parallel
from pylamp_amd._context import Context
nx = [65, 65]
grid = [np.linspace(0, 660e3, nx[0]), np.linspace(0, 660e3, nx[1])]
ctx = Context(nx, grid, attach_dist=False)
comm = parallel.Comm(ctx)
print("device_mode before:", comm.device_mode)
comm._agree_device_mode()
print("device_mode after the self-test:", comm.device_mode, "errors:", comm.errors)
dist.destroy_process_group()
