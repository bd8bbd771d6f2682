import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x=torch.arange(8,dtype=torch.float32,device=dev); out=torch.empty(8,device=dev)
dist.all_gather_into_tensor(out,x); dist.barrier(); torch.cuda.synchronize()
print("ok", out.tolist(), torch.cuda.nccl.version())
dist.destroy_process_group()
