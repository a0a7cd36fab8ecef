import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29611")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
t = torch.ones(64, device="cuda")
for _ in range(20): dist.broadcast(t, 0)
torch.cuda.synchronize()
for name, fn in (("broadcast", lambda: dist.broadcast(t, 0)), ("all_reduce", lambda: dist.all_reduce(t))):
    ts=[]
    for _ in range(50):
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    ts.sort(); print(name, "1-rank nccl: median %.1f us min %.1f us" % (ts[25]*1e6, ts[0]*1e6))
# back-to-back 100 broadcasts
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): dist.broadcast(t, 0)
torch.cuda.synchronize(); print("100 back-to-back broadcasts: %.1f us each" % ((time.perf_counter()-t0)*1e4))
dist.destroy_process_group()
