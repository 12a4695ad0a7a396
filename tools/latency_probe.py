#!/usr/bin/env python
"""Small-shape latency of the DSP tails (the real-time / streaming use: B = 1, a fraction of a second per call): our launches
as they are, the same step replayed from a captured HIP graph (torch.cuda.CUDAGraph: the C ABI launches on the capturing
stream and never allocates or synchronises, so it can be captured as is), and the reference's op chain under PyTorch-ROCm."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ddsp_svc_amd import synth

dev = torch.device("cuda:0")
SR, HOP = 44100, 512


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for kind in ("combsub", "sins", "combsubsuperfast"):
    for B, seconds in ((1, 0.25), (1, 1.16), (4, 2.0)):      # F = 22, 100, 173
        F = int(seconds * SR) // HOP + 1
        step, inp = bench.build_step(kind, B, F, 256, dev, seed=7)
        us = timeit(step)
        # one synchronised call: launch-to-result latency
        torch.cuda.synchronize()
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        lat_us = sorted(lat)[len(lat) // 2] * 1e6
        line = "%-17s B=%d %.2f s (F=%d): %.1f us per step back to back, %.1f us call-to-result" % (kind, B, seconds, F, us, lat_us)
        try:
            os.environ.setdefault("DDSP_HIP_ONE_STREAM", "1")
            graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(graph):
                out = step()
            ref = step()
            graph.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))
            gus = timeit(graph.replay)
            lat = []
            for _ in range(50):
                t0 = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            line += "; graph replay %.1f us back to back, %.1f us call-to-result, same bits %s" % (gus, sorted(lat)[len(lat) // 2] * 1e6, same)
        except Exception as e:
            line += "; graph capture failed: %s: %s" % (type(e).__name__, str(e)[:120])
        print(line, flush=True)
        if kind == "combsubsuperfast":                           # the allocation-free session of the model gui.py runs
            f0, (hm, hp, nm, nph), gz, w = inp["f0"], inp["ctrls"], inp["noise"], inp["window"]
            sess = synth.StreamingCombSubSuperFast(B, F, w, SR, HOP, dev)

            def fstep():
                sess.source(f0)
                return sess.synth(f0, hm, hp, nm, nph, gz)
            same = bool(torch.equal(fstep(), step()))
            us2 = timeit(fstep)
            torch.cuda.synchronize()
            lat = []
            for _ in range(50):
                t0 = time.perf_counter()
                fstep()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            print("%-17s B=%d %.2f s (F=%d): StreamingCombSubSuperFast session %.1f us back to back, %.1f us call-to-result, same bits %s"
                  % (kind, B, seconds, F, us2, sorted(lat)[len(lat) // 2] * 1e6, same), flush=True)
            lat = []
            sess.source(f0)
            for _ in range(50):
                t0 = time.perf_counter()
                sess.synth(f0, hm, hp, nm, nph, gz)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            print("%-17s B=%d %.2f s (F=%d): session, DSP tail alone %.1f us call-to-result" % (kind, B, seconds, F, sorted(lat)[len(lat) // 2] * 1e6),
                  flush=True)
        if kind == "combsub":                                    # the allocation-free session: the same two C calls, nothing else on the host
            f0, (cg, ch, cn), u = inp["f0"], inp["ctrls"], inp["noise"]
            sess = synth.StreamingCombSub(B, F, 256, 256, 256, SR, HOP, dev)

            def sstep():
                sess.phase(f0)
                return sess.synth(f0, cg, ch, cn, u)
            same = bool(torch.equal(sstep(), step()))
            us2 = timeit(sstep)
            torch.cuda.synchronize()
            lat = []
            for _ in range(50):
                t0 = time.perf_counter()
                sstep()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            print("%-17s B=%d %.2f s (F=%d): StreamingCombSub session %.1f us back to back, %.1f us call-to-result, same bits %s"
                  % (kind, B, seconds, F, us2, sorted(lat)[len(lat) // 2] * 1e6, same), flush=True)
            # the real-time caller's order is phase -> Unit2Control -> DSP tail (vocoder.py:822-862): what it waits for after
            # its network is the TAIL alone (three dependent launches); the phase call precedes the network
            sess.phase(f0)
            parts = {}
            for name, fn in (("tail", lambda: sess.synth(f0, cg, ch, cn, u)), ("phase", lambda: sess.phase(f0))):
                torch.cuda.synchronize()
                lat = []
                for _ in range(50):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t0)
                parts[name] = (timeit(fn), sorted(lat)[len(lat) // 2] * 1e6)
            print("%-17s B=%d %.2f s (F=%d): session, DSP tail alone %.1f us back to back, %.1f us call-to-result; phase alone %.1f / %.1f"
                  % (kind, B, seconds, F, parts["tail"][0], parts["tail"][1], parts["phase"][0], parts["phase"][1]), flush=True)
