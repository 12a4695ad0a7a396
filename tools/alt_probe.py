import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ddsp_svc_amd import _ffi, core
import bench
dev = torch.device("cuda:0")
n, N, B, F, HOP = 256, 510, 32, 862, 512
L = _ffi.lib(); st = torch.cuda.current_stream().cuda_stream
tab = core.ir_table(n, dev)
f0, ctrls, noise = bench.make_inputs("combsub", B, F, (n, n, n), dev, 1234)
rows = B * F
tp = torch.empty(rows, N, device=dev); y = torch.empty(B, F * HOP, device=dev)
hw = (1.5 * 44100.0) / (f0.reshape(-1) + 1e-3)
c = ctrls[1]; ld = c.stride(1)
def gemm(mode): _ffi.check(L.ddsp_hip_impulse_response(c.data_ptr(), ld, None, 0, 1, 1.0, mode, hw.data_ptr(), rows, n, tab.data_ptr(), tp.data_ptr(), st))
def fir(): _ffi.check(L.ddsp_hip_fft_convolve(noise.data_ptr(), 0, tp.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, 5, st))
def timed(seq, reps=10):
    for f in seq: f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        row = []
        for f in seq:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); row.append((e0, e1))
        evs.append(row)
    torch.cuda.synchronize()
    return [round(sorted(r[i][0].elapsed_time(r[i][1]) for r in evs)[reps // 2] * 1e3, 1) for i in range(len(seq))]
print("dyn alone", timed([lambda: gemm(2)]))
print("hann alone", timed([lambda: gemm(1)]))
print("fir, dyn", timed([fir, lambda: gemm(2)]))
print("fir, hann", timed([fir, lambda: gemm(1)]))
print("fir, fir, dyn, dyn", timed([fir, fir, lambda: gemm(2), lambda: gemm(2)]))
