import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n = 64 * 300 * 6 + 64
loc = torch.zeros(n, device="cuda"); out = torch.zeros(n, device="cuda")
for _ in range(5): dist.all_gather_into_tensor(out, loc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): dist.all_gather_into_tensor(out, loc)
torch.cuda.synchronize()
print("sync all_gather us/call", (time.perf_counter() - t0) / 100 * 1e6)
t0 = time.perf_counter()
hs = [dist.all_gather_into_tensor(out, loc, async_op=True) for _ in range(100)]
for h in hs: h.wait()
torch.cuda.synchronize()
print("async all_gather us/call", (time.perf_counter() - t0) / 100 * 1e6)
x = torch.zeros(1 << 20, device="cuda")
t0 = time.perf_counter()
for _ in range(100):
    x.add_(1.0); dist.all_gather_into_tensor(out, loc)
torch.cuda.synchronize()
print("kernel + all_gather us/iter", (time.perf_counter() - t0) / 100 * 1e6)
# pipelined: gather of iteration i overlaps the kernel of iteration i+1 (double buffer, wait two iterations later)
locs = [torch.zeros(n, device="cuda") for _ in range(2)]; outs = [torch.zeros(n, device="cuda") for _ in range(2)]
hs = [None, None]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    k = i & 1
    if hs[k] is not None: hs[k].wait()
    x.add_(1.0); locs[k].add_(1.0)
    hs[k] = dist.all_gather_into_tensor(outs[k], locs[k], async_op=True)
for h in hs: h.wait()
torch.cuda.synchronize()
print("pipelined kernel + async all_gather us/iter", (time.perf_counter() - t0) / 100 * 1e6)
# plain cross-stream event ping-pong without NCCL
s2 = torch.cuda.Stream(); cur = torch.cuda.current_stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    x.add_(1.0)
    e = torch.cuda.Event(); e.record(cur); s2.wait_event(e)
    with torch.cuda.stream(s2): out.copy_(loc)
    e2 = torch.cuda.Event(); e2.record(s2); cur.wait_event(e2)
torch.cuda.synchronize()
print("kernel + side-stream copy ping-pong us/iter", (time.perf_counter() - t0) / 100 * 1e6)
dist.destroy_process_group()
