import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=[torch.zeros(1,dtype=torch.int64,device="cuda")]
dist.all_gather(t, torch.tensor([7],dtype=torch.int64,device="cuda"))
dist.barrier(); x=torch.tensor([1.5],dtype=torch.float64,device="cuda"); dist.all_reduce(x, op=dist.ReduceOp.MAX)
print("nccl world=1 ok", t[0].item(), x.item())
dist.destroy_process_group()
