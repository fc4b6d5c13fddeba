import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29561")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from online_lang_splatting_amd.rccl_direct import DirectComm
t0=time.time(); dc = DirectComm.from_process_group(); print("init", time.time()-t0)
x = torch.arange(16, dtype=torch.float32, device="cuda")
dc.all_reduce(x, "sum"); torch.cuda.synchronize(); print(x[:4])
dc.destroy(); dist.destroy_process_group(); print("ok")
