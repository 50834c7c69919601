"""Does an initialised RCCL communicator slow down cross-stream fork/join of unrelated kernels?"""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
mode = sys.argv[1]
def pingpong(tag):
    x = torch.zeros(1 << 22, device="cuda"); y = torch.zeros(1 << 22, device="cuda")
    s2 = torch.cuda.Stream(); cur = torch.cuda.current_stream()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200):
            e = torch.cuda.Event(); e.record(cur); s2.wait_event(e)
            x.add_(1.0)
            with torch.cuda.stream(s2): y.add_(1.0)
            e2 = torch.cuda.Event(); e2.record(s2); cur.wait_event(e2)
        torch.cuda.synchronize()
    print(tag, "fork/join us/iter", (time.perf_counter() - t0) / 200 * 1e6, flush=True)
pingpong("before init")
if mode == "eager":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
elif mode == "lazy":
    dist.init_process_group("nccl")
pingpong("after init (%s)" % mode)
if mode == "lazy":
    t = torch.zeros(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    pingpong("after first collective")
dist.destroy_process_group()
pingpong("after destroy")
