"""Why is the B = 32 step ~10 % slower in a process that holds an RCCL communicator (0.35 ms against 0.317: every cfg-4 / --gather
line since round 2 shows it, and every rank of a --gpus N run holds one)?  One configuration per process:
    python tools/pg_probe.py <none|nccl|nccl_nomon|gloo|nccl_destroyed> [graph]
prints the step time (200 steps between fences, wall and HIP events), the host's enqueue time per step (20 steps, no wait) and,
with `graph`, the same step replayed as one captured graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
graph = "graph" in sys.argv[2:]
B = int(os.environ.get("PROBE_B", "32"))
if mode == "nccl_nomon":
    os.environ["TORCH_NCCL_ENABLE_MONITORING"] = "0"
    os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
_keep = []
if mode.startswith("streams"):                       # idle streams (and events) of other priorities, no communicator: is it their mere existence?
    n = int(mode[7:] or 8)
    for i in range(n):
        st_ = torch.cuda.Stream(priority=-1 if i % 2 else 0)
        ev_ = torch.cuda.Event()
        with torch.cuda.stream(st_):
            torch.zeros(1, device=dev)
            ev_.record()
        _keep.append((st_, ev_))
    torch.cuda.synchronize()
elif mode == "pinned":                               # a pinned host allocation + a mapped one, as a communicator holds
    _keep.append(torch.empty(64 << 20, dtype=torch.uint8).pin_memory())
elif mode != "none":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    if mode == "gloo":
        dist.init_process_group("gloo", rank=0, world_size=1)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        if mode == "nccl_destroyed":
            dist.destroy_process_group()
F = 862
if os.environ.get("PROBE_MAIN_PRIORITY"):            # the CALLER's stream in another priority class (the second stream stays normal)
    _main = torch.cuda.Stream(priority=int(os.environ["PROBE_MAIN_PRIORITY"]))
    torch.cuda.set_stream(_main)
step, _ = bench.build_step("combsub", B, F, 256, dev, seed=1)
bench.prewarm(step, 0.5)
if graph:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = step()
    fn = gr.replay
else:
    fn = step
for _ in range(20):
    fn()
torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res.append(((time.perf_counter() - t0) / 200 * 1e3, e0.elapsed_time(e1) / 200))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fn()
host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print("%-16s graph=%d B=%d  ms/step wall %s  events %s  host enqueue %.4f ms/step  threads %d"
      % (mode, graph, B, ["%.4f" % r[0] for r in res], ["%.4f" % r[1] for r in res], host, len(os.listdir("/proc/self/task"))))
if mode in ("nccl", "nccl_nomon", "gloo"):
    import torch.distributed as dist
    dist.destroy_process_group()
