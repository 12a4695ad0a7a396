#!/usr/bin/env python
"""Headline benchmark: audio samples/s of the CombSub DSP path (BASELINE.json cfg 2) on N MI355X.

A step = one pass of the hot path over one resident batch: HOT-1 (phase scan, phase_frames) +
HOT-2 (combtooth, 3x tap synthesis, 3x time-varying FIR, mix) for B utterances of 10 s at
44.1 kHz / hop 512 with 256/256/256 magnitude bins.  Inputs (f0, raw controls ~N(0,1), uniform
noise) are already in HBM; only ``signal`` is written (the (harmonic, noise) tuple every reference
caller discards is not materialised).  Multi-GPU: one process per GPU, utterances sharded, no
data-path collective ("weak" scaling: B per GPU fixed).

``--gpus N`` with N > 1 and no RANK in the environment re-executes itself under ``torch.distributed.run`` with N ranks
(one per GPU, backend nccl = RCCL); under an external launcher it checks WORLD_SIZE == N.  It never falls back to fewer
ranks: fewer than N visible devices is an error.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline               the dominant kernel (the time-varying FIR, k_fir_blk6) INSIDE the step at the clocks' steady state: a
                         rocprofv3 --kernel-trace pass of this run (900 steps, the launches past the first third); frac = SURVEY 8-d
                         bytes of what one launch processes / that duration / 8 TB/s; bound "valu" (valu_frac beside it); frac_alone
                         = the kernel timed alone with HIP events; step_hbm_frac / step_traffic_ratio = the whole step's
  roofline_step_traffic  PMC HBM bytes of one whole step over its algorithmic bytes: measured IN this run when rocprofv3 is
                         on the box (two --pmc passes over ``bench.py --only-steps``, ``traffic_source: "live"``), otherwise the
                         newest committed profiles/*_hbm_traffic.json (``traffic_source: "committed"``)
  also                   (N = 1) the other single-GPU BASELINE configs attested by the same command: the Sins cfg-3 step with its
                         dominant kernel's roofline, the as-shipped CombSubSuperFast model (informative), train_combsub = forward +
                         backward of the CombSub DSP behind a gradient gate against the oracle's adjoint, and the reference's loss
  cfg4                   (the default N = 1 command, N = 8, or --cfg4) BASELINE cfg 4's per-GPU shape: 64 utterances per GPU,
                         samples/s without and with the RCCL gather, interleaved rounds (a 1-rank communicator at N = 1)
  parity_vs_oracle       the gate every line is printed behind: two utterances of the TIMED output against the oracle
                         (<= 1e-4 RMS abs, <= 1e-5 rel) -- a miss is SystemExit, not a field
  hip_hw_queues          GPU_MAX_HW_QUEUES of the process (set to 8 here before the HIP runtime loads: see below; since round 6 the
                         default launch layout has no second stream and does not depend on it)
  ms_per_step_events     the same K timed steps measured with HIP events on the launch stream, beside the wall clock
  cpu_baseline           the numpy oracle (oracle/ddsp_oracle.py, a port of the reference algorithm) timed
                         on this host's cores over a bounded sample of the same workload (N=1 only); kind "reference" -- the
                         UNMODIFIED reference module itself, with the 1e-4 parity gate -- when a reference checkout
                         travelled with the run (DDSP_REFERENCE_PATH, tools/with_reference.sh; the port then moves to
                         cpu_baseline_port)
  cpu_baseline_aten_chain  the reference's op chain walked with torch CPU operators (oracle/aten_chain.py), all cores;
                           .same_chain_on_gpu: the same chain with its tensors on the GPU (PyTorch-ROCm), full workload
  value_module_mode      the drop-in module with a stand-in Unit2Control producing the controls on the GPU (control mode (i))
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BEFORE the HIP runtime is loaded (import torch): eight hardware queues instead of the runtime's default four.  HIP maps
# streams onto its hardware queues round robin; a process that holds an RCCL communicator (every rank of --gpus N, the cfg-4
# line) has created ~7 streams before the synthesiser makes its second one, which then lands on the SAME hardware queue as the
# caller's stream: the two-stream layout runs in one queue and the B = 32 step takes 0.359 ms instead of 0.325 (same box,
# profiles/r05_v5_pg_fix.txt; with eight queues 0.326 with and without the communicator; every 1-rank "gather" line of rounds
# 2 - 4 carried this +10 %).  A user's setting wins.  INTEGRATION.md, "processes that hold an RCCL communicator".
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

SR, HOP = 44100, 512


FAST_MODELS = ("combsubfast", "combsubsuperfast")


class _HipRuntime:
    """The device layer the rank logic below talks to: PyTorch-ROCm's HIP runtime and RCCL (``nccl``).  bench.py itself has
    no other: without an MI355X it refuses to run.  tests/test_bench_ranks.py swaps in an emulator-backed object (CPU
    tensors, ``gloo``) to drive the SAME setup_ranks / fence / reduce-max / rank-0-emit code under two ranks on a host
    without GPUs, so that the first real ``--gpus 8`` run does not die in a branch that never executed."""
    backend = "nccl"

    def available(self):
        return torch.cuda.is_available()

    def device_count(self):
        return torch.cuda.device_count() if torch.cuda.is_available() else 0

    def device(self, local_rank):
        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    def synchronize(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def empty_cache(self):
        torch.cuda.empty_cache()

    def init_process_group(self, rank, world, device):
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)


RT = _HipRuntime()


def prewarm(step, seconds):
    """Bring the GPU to its sustained clocks before anything is timed: after idling the first ~50-100 ms of work run
    7-8 % slower (measured: the same fused step takes 0.64 ms in the first 20 ms of a process and 0.596 ms later).
    Not counted as warm-up or timed steps; the W warm-up steps and K timed steps follow as the contract says."""
    if seconds <= 0:
        return 0
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        RT.synchronize()
        n += 20
    return n


def model_sizes(kind, bins):
    """control channels per split (ddsp/vocoder.py:549-554, :804-808, :728-732, :631-636)"""
    if kind == "combsubfast":
        return (HOP + 1,) * 3
    if kind == "combsubsuperfast":
        return (1025,) * 4
    return (bins,) * 3


def synthetic_f0(B, F, seed):
    """vibrato + random-walk drift curves clamped to [65, 800] Hz, strictly positive, some utterances above
    259 Hz (the dynamic-window clamp of core.py:245 is exercised); ``[B,F,1]`` float32"""
    rng = np.random.default_rng(seed)
    base = rng.uniform(100.0, 400.0, size=(B, 1))
    depth = rng.uniform(0.0, 1.0, size=(B, 1))
    phi = rng.uniform(0.0, 2.0 * np.pi, size=(B, 1))
    drift = np.cumsum(rng.normal(0.0, 0.05, size=(B, F)), axis=1)
    t = np.arange(F)[None, :] * HOP / SR
    semitones = depth * np.sin(2.0 * np.pi * 5.5 * t + phi) + 0.5 * drift
    return np.clip(base * 2.0 ** (semitones / 12.0), 65.0, 800.0).astype(np.float32)[:, :, None]


def make_inputs(kind, B, F, sizes, device, seed):
    f0 = torch.from_numpy(synthetic_f0(B, F, seed)).to(device)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    ctrl = torch.randn(B, F, sum(sizes), generator=g).to(device)      # one tensor, split into strided views
    ctrls = torch.split(ctrl, list(sizes), dim=-1)
    if kind == "combsubsuperfast":
        noise = torch.randn(B, F * HOP, generator=g).to(device)       # randn_like (vocoder.py:687)
    else:
        noise = (torch.rand(B, F * HOP, generator=g) * 2 - 1).to(device)
    return f0, ctrls, noise


def _cpu_worker_init():
    try:                                            # one thread per worker: the pool already uses every core
        import threadpoolctl
        _cpu_worker_init.limiter = threadpoolctl.threadpool_limits(1)
    except Exception:
        pass


def _cpu_worker(args):
    seed, F, kind, sizes = args
    import numpy as _np
    from oracle import ddsp_oracle as O
    f0 = O.synth_f0(1, F, SR, HOP, seed=seed)
    cs = O.synth_controls(1, F, sizes, seed=seed + 1)
    nz = O.synth_noise(1, F * HOP, seed=seed + 2)
    if kind == "combsub":
        r = O.combsub_dsp(f0, cs[0], cs[1], cs[2], nz, SR, HOP)
    elif kind == "combsubfast":
        r = O.combsubfast_dsp(f0, cs[0], cs[1], cs[2], nz, SR, HOP)
    elif kind == "combsubsuperfast":
        r = O.combsubsuperfast_dsp(f0, cs[0], cs[1], cs[2], cs[3], O.synth_gauss(1, F * HOP, seed=seed + 2), SR, HOP)
    else:
        r = O.sins_dsp(f0, cs[0], cs[1], cs[2], nz, SR, HOP)
    return float(_np.abs(r["signal"]).max())


def hbm_traffic(kernel, model=None):
    """HBM bytes per launch of ``kernel`` from the newest committed PMC summary of ``model``'s step (tools/gpu_traffic.sh:
    separate FETCH_SIZE / WRITE_SIZE passes over this same workload, gfx950 read correction applied), or None."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    def version(name):                                   # r01_v10_... after r01_v9_...: numeric, not lexicographic
        import re
        return [int(x) for x in re.findall(r"\d+", name)]
    for name in sorted(os.listdir(pdir), key=version) if os.path.isdir(pdir) else []:
        if name.endswith("_hbm_traffic.json"):
            try:
                d = json.load(open(os.path.join(pdir, name)))
            except Exception:
                continue
            if model is not None and d.get("__step__", {}).get("model", model) != model:
                continue
            for k, v in d.items():
                if k.startswith(kernel):
                    best = {"bytes": v["hbm_bytes"], "read": v["read_bytes"], "write": v["write_bytes"], "source": "profiles/" + name}
    return best


def live_traffic(model, extra_args=()):
    """HBM bytes per kernel launch and per step of ``model``, measured now: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
    separate passes (they do not fit one; kernel trace only) over ``bench.py --model M --only-steps --steps 3 --warmup 1``,
    FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM), both in KiB.  The same
    arithmetic as tools/gpu_traffic.sh + tools/traffic_summary.py.  None if rocprofv3 is absent or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    steps, warm = 3, 1
    raw = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ddsp_traffic_", dir="/tmp")
        try:
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "t", "--", sys.executable,
                   os.path.abspath(__file__), "--model", model, "--only-steps", "--steps", str(steps), "--warmup", str(warm),
                   *extra_args]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            c = sqlite3.connect(dbs[0])
            rows = c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? "
                             "group by kernel_name", (counter,)).fetchall()
            c.close()
            for name, v, cnt in rows:
                if "ddsp::" not in name:
                    continue
                key = name.split("(")[0].split("ddsp::")[-1].strip()
                raw.setdefault(key, {})[counter] = float(v)
                raw[key]["launches"] = int(cnt)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    res = {}
    for k, d in raw.items():
        rd, wr = d.get("FETCH_SIZE", 0.0) * 2048.0, d.get("WRITE_SIZE", 0.0) * 1024.0
        res[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "launches_sampled": d.get("launches")}
    per_step = {k: v["launches_sampled"] / float(steps + warm) for k, v in res.items() if k != "k_ir_table"}
    res["__step__"] = {"model": model, "steps_profiled": steps + warm,
                       "hbm_bytes": sum(res[k]["hbm_bytes"] * n for k, n in per_step.items()),
                       "read_bytes": sum(res[k]["read_bytes"] * n for k, n in per_step.items()),
                       "write_bytes": sum(res[k]["write_bytes"] * n for k, n in per_step.items()),
                       "launches_per_step": per_step, "seconds": time.perf_counter() - t0}
    return res


def in_step_kernel_times(model, extra_args=(), steps=900, stats_out=None):
    """Average duration of every ddsp:: kernel INSIDE the step at the clocks' steady state: one ``rocprofv3 --kernel-trace`` pass
    (no counters) over ``bench.py --model M --only-steps --steps 900``, the last TWO THIRDS of each kernel's launches (a process's first
    ~100 steps are a transient -- a filter launch goes 77 -> 100 -> 72 us while the clocks settle, profiles/r06_v6_*, r06_v7_* --
    and every trace of rounds 2 - 5 was 12 - 24 steps long: the "in-step penalty" those rounds chased was that transient).
    Returns {kernel: {"avg_us", "launches_per_step", "avg_us_all"}} or None.  ``stats_out``: also write the
    ``rocprofv3 --stats``-style table of the WHOLE trace there (what profiles/*_kernel_stats.csv holds)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    warm = 10
    d = tempfile.mkdtemp(prefix="ddsp_trace_", dir="/tmp")
    try:
        cmd = [rocprof, "--kernel-trace", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--model", model,
               "--only-steps", "--steps", str(steps), "--warmup", str(warm), *extra_args]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None
        c = sqlite3.connect(dbs[0])
        rows = c.execute("select name, start, end from kernels order by start").fetchall()
        c.close()
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)
    per = {}
    for name, s0, e0 in rows:
        if "ddsp::" not in name:
            continue
        per.setdefault(name.split("(")[0].split("ddsp::")[-1].strip(), []).append((e0 - s0) / 1e3)
    out, steady = {}, {}
    for k, v in per.items():
        if k == "k_ir_table" or len(v) < steps:
            continue
        tail = v[len(v) // 3:]                                  # past the clocks' transient: the last two thirds of the launches
        steady[k] = tail
        out[k] = {"avg_us": sum(tail) / len(tail), "launches_per_step": round(len(v) / float(steps + warm), 3),
                  "avg_us_all": sum(v) / len(v), "launches_traced": len(v), "launches_averaged": len(tail)}
    if stats_out:
        tot = sum(sum(v) for v in steady.values()) or 1.0
        lines = ["# rocprofv3 --kernel-trace of `bench.py --model %s --only-steps --steps %d`: per kernel, the launches past the clocks' "
                 "transient (the last two thirds); avg_us_whole_trace = all of them" % (model, steps),
                 "kernel,calls,total_us,avg_us,min_us,max_us,percent,avg_us_whole_trace"]
        for k, v in sorted(steady.items(), key=lambda kv: -sum(kv[1])):
            lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f,%.2f' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot,
                                                                  sum(per[k]) / len(per[k])))
        with open(stats_out, "w") as f:
            f.write("\n".join(lines) + "\n")
    return out or None


def traffic_of(kernel, model, live):
    """(per-launch traffic of ``kernel``, whole-step traffic, source) from the live measurement if there is one, else from
    the newest committed summary"""
    if live:
        k = next((v for name, v in live.items() if name.startswith(kernel)), None)
        kt = None if k is None else {"bytes": k["hbm_bytes"], "read": k["read_bytes"], "write": k["write_bytes"],
                                     "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"}
        return kt, dict(live["__step__"], source="live"), "live"
    return hbm_traffic(kernel, model), step_traffic(model), "committed"


def build_step(model, B, F, n, device, seed, fir_impl=0):
    """(step, inputs) of ``model``: HOT-1 + HOT-2 from resident f0 / raw controls / noise, signal only"""
    from ddsp_svc_amd import synth
    sizes = model_sizes(model, n)
    f0, ctrls, noise = make_inputs(model, B, F, sizes, device, seed=seed)
    win = 2048 if model == "combsubsuperfast" else 2 * HOP
    window = torch.hann_window(win, device=device)
    if model == "combsubfast":
        window = torch.sqrt(window)

    def step():
        if model == "combsubsuperfast":
            fs = synth.fast_source(f0, SR, HOP)
            return synth.combsubsuperfast_synth(f0, fs, ctrls[0], ctrls[1], ctrls[2], ctrls[3], noise, window, SR, HOP)
        st = synth.phase(f0, SR, HOP)
        if model == "combsubfast":
            return synth.combsubfast_synth(f0, st, ctrls[0], ctrls[1], ctrls[2], noise, window, SR, HOP)
        if model == "combsub":
            return synth.combsub_synth(f0, st, ctrls[0], ctrls[1], ctrls[2], noise, SR, HOP,
                                       want_components=False, fir_impl=fir_impl)[0]
        return synth.sins_synth(f0, st, ctrls[0], ctrls[1], ctrls[2], noise, SR, HOP,
                                want_components=False, fir_impl=fir_impl)[0]
    return step, {"f0": f0, "ctrls": ctrls, "noise": noise, "sizes": sizes, "window": window, "win": win}


def time_steps(step, steps, warmup, fence):
    """(wall seconds, HIP-event ms per step, last output) of exactly ``steps`` steps after ``warmup`` untimed ones"""
    for _ in range(warmup):
        out = step()
    fence()
    e0, e1 = RT.event(), RT.event()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        out = step()
    e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    return elapsed, e0.elapsed_time(e1) / steps, out


def time_alone(fn, reps=20):
    """HIP-event ms per call of ``fn`` launched back to back on the current stream"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def also_lines(a, device, B, F, n, want_traffic):
    """The other single-GPU BASELINE configs, measured by the same command (N = 1): cfg 3 = the Sins step with the
    roofline of its dominant kernel (the sinusoid bank), and the as-shipped configs/combsub.yaml model CombSubSuperFast
    (informative).  Fewer steps than the headline (the whole block costs about two seconds)."""
    from ddsp_svc_amd import synth
    T = F * HOP
    steps, warm = max(10, min(a.steps, 30)), 5
    fence = torch.cuda.synchronize
    out = {}
    # ---- cfg 3: Sins, 256 harmonics / 256 / 256 bins
    step, inp = build_step("sins", B, F, n, device, seed=4321)
    prewarm(step, min(a.prewarm_seconds, 0.2))
    el, ev_ms, y = time_steps(step, steps, warm, fence)
    assert torch.isfinite(y).all()
    st = synth.phase(inp["f0"], SR, HOP)
    bank_ms = time_alone(lambda: synth.sinusoid_bank(inp["f0"], st, inp["ctrls"][0], SR, HOP))
    bank_bytes = (4.0 + 4.0 * n / HOP) * B * T                 # amplitude controls in, exciter out
    alg = (8.0 + 4.0 * (sum(inp["sizes"]) + 1) / HOP) * B * T
    live = live_traffic("sins") if want_traffic else None
    kt, stt, src = traffic_of("k_sins_bank3", "sins", live) if (B, F, n) == (32, 862, 256) else (None, None, None)
    ms = el / steps * 1e3
    out["sins"] = {
        "metric": "audio samples/sec, Sins 44.1kHz 256-harm hop512", "value": B * T * steps / el, "unit": "samples/s",
        "steps": steps, "warmup": warm, "ms_per_step": ms, "ms_per_step_events": ev_ms, "dtype": "f32",
        "config": {"workload": "sins B=%d x %.0f s utterances (F=%d, T=%d), %d harmonics, n_mag %d/%d, DSP path from resident "
                               "f0 / raw controls / uniform noise, signal only (BASELINE cfg 3)" % (B, a.seconds, F, T, n, n, n)},
        "roofline": {"kernel": "k_sins_bank3", "bound": "valu", "achieved": bank_bytes / (bank_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": bank_bytes / (bank_ms * 1e-3) / 1e9 / 8000.0,
                     "traffic": kt["bytes"] if kt else None, "traffic_source": src,
                     "algorithmic_bytes_per_launch": bank_bytes, "avg_ms": bank_ms, "launches_per_step": 1,
                     "note": "issue-bound, not HBM-bound: %.2f G sine terms per launch (DESIGN.md, sinusoid bank)" % (n * B * T / 1e9)},
        "roofline_step_hbm": {"algorithmic_bytes_per_step": alg, "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0,
                              "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / 8000.0},
        "roofline_step_traffic": None if not stt else {"pmc_bytes_per_step": stt["hbm_bytes"], "ratio": stt["hbm_bytes"] / alg,
                                                       "source": stt["source"]}}
    del step, inp, y
    # ---- the as-shipped configs/combsub.yaml model
    step, inp = build_step("combsubsuperfast", B, F, n, device, seed=4322)
    prewarm(step, min(a.prewarm_seconds, 0.2))
    el, ev_ms, y = time_steps(step, steps, warm, fence)
    assert torch.isfinite(y).all()
    exc = torch.randn(B, T, device=device)
    c = inp["ctrls"]
    k_ms = time_alone(lambda: synth.stft_filter(exc, inp["noise"], c[0], c[1], c[2], c[3], inp["window"], HOP,
                                                pad_reflect=True, normalize=True))
    k_bytes = (12.0 + 4.0 * sum(inp["sizes"]) / HOP) * B * T
    ms = el / steps * 1e3
    out["combsubsuperfast"] = {
        "metric": "audio samples/sec, CombSubSuperFast 44.1kHz win2048 hop512", "value": B * T * steps / el,
        "unit": "samples/s", "steps": steps, "warmup": warm, "ms_per_step": ms, "ms_per_step_events": ev_ms, "dtype": "f32",
        "config": {"workload": "combsubsuperfast B=%d x %.0f s (F=%d, T=%d), 4 x 1025 control bins (configs/combsub.yaml as "
                               "shipped; informative)" % (B, a.seconds, F, T)},
        "roofline": {"kernel": "k_stft_filter<4>", "bound": "hbm", "achieved": k_bytes / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": k_bytes / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": k_bytes, "avg_ms": k_ms, "launches_per_step": 1}}
    del step, inp, y, exc, c
    # ---- a training step of the DSP (solver.py:93-103 calls the same forward with gradients): forward + backward of the CombSub
    # tail w.r.t. its three raw controls, cotangent = a fixed random [B, T]; GATED like the headline: utterance 0's three gradients
    # of the last timed step against the oracle's float64 adjoint (oracle.combsub_dsp_backward, pinned to the reference's own
    # autograd by tests/test_oracle_golden.py::test_combsub_tail_adjoint) <= 1e-4 relative each, else no line
    out["train_combsub"] = train_line(a, device, B, F, n, steps, warm)
    # ---- the training loss of the reference (train.py:69): RSSLoss, 4 scales of arbitrary (here: three prime) size
    from ddsp_svc_amd import loss as L
    g = torch.Generator(device="cpu").manual_seed(99)
    xt = (torch.randn(B, T, generator=g) * 0.1).to(device)
    xp = (xt * 0.9 + 0.02 * torch.randn(B, T, generator=g).to(device)).requires_grad_(True)
    sizes = [1153, 397, 2011, 768]
    rss, drawn, real_randint = L.RSSLoss(256, 2048, len(sizes), device=device), torch.tensor(sizes), torch.randint

    def loss_step():
        torch.randint = lambda *a_, **k_: drawn                      # loss.py:47's draw, pinned
        try:
            value = rss(xp, xt)
        finally:
            torch.randint = real_randint
        return torch.autograd.grad(value, xp)[0]
    prewarm(loss_step, min(a.prewarm_seconds, 0.2))
    el, ev_ms, y = time_steps(loss_step, steps, warm, fence)
    assert torch.isfinite(y).all()
    out["rssloss"] = {
        "metric": "audio samples/sec, RSSLoss forward+backward 44.1kHz 4 scales", "value": B * T * steps / el,
        "unit": "samples/s", "steps": steps, "warmup": warm, "ms_per_step": el / steps * 1e3, "ms_per_step_events": ev_ms,
        "dtype": "f32",
        "config": {"workload": "RSSLoss(x_pred, x_true) + d/dx_pred, B=%d x %.0f s, transform sizes %s (hop = size), STFT inside "
                               "the loss kernels (chirp-z, csrc/loss_czt.hip); `--model rssloss` gives the full line"
                               % (B, a.seconds, sizes)}}
    return out


def train_line(a, device, B, F, n, steps, warm):
    """``also.train_combsub``: forward + backward of the CombSub DSP (torch.autograd.grad through synth.combsub_synth) at the
    headline shape, behind a gradient parity gate against the oracle."""
    from ddsp_svc_amd import synth
    from oracle import ddsp_oracle as O
    T = F * HOP
    f0, ctrls, noise = make_inputs("combsub", B, F, model_sizes("combsub", n), device, seed=777)
    c = [x.clone().requires_grad_(True) for x in ctrls]
    g = torch.Generator(device="cpu").manual_seed(778)
    R = torch.randn(B, T, generator=g).to(device)

    def step():
        st = synth.phase(f0, SR, HOP)
        sig = synth.combsub_synth(f0, st, c[0], c[1], c[2], noise, SR, HOP)[0]
        # the cotangent R is handed to autograd as a loss would hand it (the gradient of sum(signal * R), without timing torch's own
        # multiply, reduction and two fills for that scalar: ~76 us that are neither the product nor the reference's)
        return torch.autograd.grad(sig, c, grad_outputs=R)
    prewarm(step, min(a.prewarm_seconds, 0.3))
    el, ev_ms, grads = time_steps(step, steps, warm, torch.cuda.synchronize)
    t0 = time.perf_counter()
    host = lambda t: np.ascontiguousarray(t[0:1].detach().cpu().numpy())
    want = O.combsub_dsp_backward(host(R), host(f0), host(c[0]), host(c[1]), host(c[2]), host(noise), SR, HOP)
    errs = {}
    for gk, k in zip(grads, ("group_delay", "harmonic_magnitude", "noise_magnitude")):
        got = gk[0:1].detach().cpu().numpy().astype(np.float64)
        errs[k] = float(np.sqrt(np.mean((got - want[k]) ** 2)) / max(float(np.sqrt(np.mean(want[k] ** 2))), 1e-30))
    rec = {"rows": [0], "rel_rms": errs, "bar_rel": 1e-4, "seconds": time.perf_counter() - t0,
           "oracle": "oracle/ddsp_oracle.py combsub_dsp_backward (float64 adjoint, pinned to the reference's autograd fixture)"}
    if not all(np.isfinite(v) and v <= 1e-4 for v in errs.values()):
        raise SystemExit("bench.py: the timed training step's gradients are NOT the reference's: %s" % json.dumps(rec))
    return {"metric": "audio samples/sec, CombSub DSP forward+backward 44.1kHz 256-harm hop512", "value": B * T * steps / el,
            "unit": "samples/s", "steps": steps, "warmup": warm, "ms_per_step": el / steps * 1e3, "ms_per_step_events": ev_ms,
            "dtype": "f32", "parity_vs_oracle": rec,
            "config": {"workload": "combsub B=%d x %.0f s (F=%d, T=%d), n_mag %d/%d/%d: phase + DSP tail forward, then "
                                   "torch.autograd.grad of signal w.r.t. the three raw controls with the cotangent R as grad_outputs "
                                   "(= the gradient of sum(signal * R); solver.py:93-103's use of the path, without the network "
                                   "and the loss)" % (B, a.seconds, F, T, n, n, n)}}


def cfg4_line(a, rank, world, device, F, n, comm):
    """BASELINE cfg 4 (combsub, 512 utterances sharded over 8 GPUs = 64 per GPU, RCCL gather over xGMI): samples/s of the
    sharded synthesis alone (A) and with the gather of every step's waveforms to rank 0 (B), max over ranks.  Measured as
    INTERLEAVED rounds A B A B .. after a clock ramp-up and warm-up steps of both forms, wall clock (between fences) beside HIP
    events, at least 50 steps of each: a drift of the box moves both columns alike, and the reported figures are medians over the
    rounds with the paired difference B - A beside them (round 4 timed 20 cold steps of A, then 20 of B, and read B < A)."""
    import torch.distributed as dist
    from ddsp_svc_amd import sharding
    B4 = a.cfg4_batch
    T = F * HOP
    if not a.cfg4_keep_cache and hasattr(RT, "empty_cache"):
        RT.empty_cache()                 # the headline's cached blocks go back to the driver: this line's buffers get fresh segments
    step, inp4 = build_step("combsub", B4, F, n, device, seed=9000 + rank, fir_impl=a.fir_impl)
    per = max(2, min(a.cfg4_round_steps, a.steps))
    rounds = 6 if a.steps >= 20 else 2
    if not dist.is_initialized() and world == 1 and not a.no_cfg4_gather:
        # the default N = 1 command times the gather too: a 1-rank communicator, the same call path (a failure to open it
        # costs the gather column, not the line)
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:                          # a free port: two benches on one host do not collide
                import socket
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            RT.init_process_group(rank, world, device)
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)
            comm["rccl_ranks"] = int(probe.item())
        except Exception as e:                                           # noqa: BLE001
            comm["cfg4_communicator_error"] = "%s: %s" % (type(e).__name__, e)
    have_pg = dist.is_initialized()

    def fence():
        if have_pg and world > 1:
            dist.barrier()
        RT.synchronize()

    def reduce_max(*vals):
        if not (have_pg and world > 1):
            return vals
        tt = torch.tensor(vals, dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in tt)

    def step_g():
        return sharding.gather_utterances(step(), B4 * world, dst=0)

    # the gather as a pipelined serving loop could issue it: step k's waveforms travel (async collective on the communicator's
    # stream) while step k + 1 is synthesised; a step's result is taken one step later; the timed region ends with the last gather
    # drained.  Measured beside the plain form because at ONE rank it LOSES (the "gather" is a 113 MB device-to-device copy whose
    # blit competes with the synthesis for the same CUs and HBM: +0.056 ms per step against +0.037 waited for in line, r06_v13);
    # over xGMI the transfer (7 x 113 MB into rank 0) is several steps long whichever way it is issued.
    pending = [None]

    def step_async():
        y = step()
        fin, _ = sharding.gather_utterances(y, B4 * world, dst=0, async_op=True)
        prev, pending[0] = pending[0], fin
        return prev() if prev is not None else y

    def fence_g():
        if pending[0] is not None:
            pending[0]()
            pending[0] = None
        fence()

    # the gather WITHOUT a copy of the rank's own shard: the receiving rank keeps the result tensor from step to step and its
    # synthesis writes its waveforms straight into its slice of it (synth.combsub_synth(signal_out=), gather_utterances(out=)):
    # nothing moves at one rank; at N ranks rank 0 saves the 113 MB device-to-device copy of its own shard
    from ddsp_svc_amd import synth as _synth
    result = [None]

    def step_in_place():
        if rank == 0 and result[0] is None:
            result[0] = torch.empty(B4 * world, T, dtype=torch.float32, device=device)
        f0, c, nzz = inp4["f0"], inp4["ctrls"], inp4["noise"]
        st = _synth.phase(f0, SR, HOP)
        mine = result[0][rank * B4:(rank + 1) * B4] if rank == 0 else None
        y = _synth.combsub_synth(f0, st, c[0], c[1], c[2], nzz, SR, HOP, want_components=False, fir_impl=a.fir_impl, signal_out=mine)[0]
        return sharding.gather_utterances(y, B4 * world, dst=0, out=result[0])
    prewarm(step, min(a.prewarm_seconds, 0.3))
    forms = [("plain", step, fence)] + ([("gather", step_g, fence), ("gather_async", step_async, fence_g)] if have_pg else [])
    if have_pg:
        failed = 0.0
        try:                                                             # a form that fails costs its column, not the line
            chk = step_in_place()
            fence()
            ref = step()                                                 # (every rank: the fences are collective)
            fence()
            if rank == 0:
                assert chk is result[0] and torch.equal(chk[:B4], ref), "in-place gather: rank 0's slice is not the step's output"
        except Exception as e:                                           # noqa: BLE001
            failed = 1.0
            comm["cfg4_in_place_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        (failed,) = reduce_max(failed)                                   # every rank times the same forms
        if failed == 0.0:
            forms.append(("gather_in_place", step_in_place, fence))
    wall = {k: [] for k, _, _ in forms}
    ev = {k: [] for k, _, _ in forms}
    for _, fn, fc in forms:                                              # warm-up of every form (allocator, communicator buffers)
        for _ in range(3):
            out = fn()
        fc()
    for _ in range(rounds):
        for k, fn, fc in forms:
            el, ev_ms, out = time_steps(fn, per, 0, fc)
            el, ev_ms = reduce_max(el, ev_ms)
            wall[k].append(el / per * 1e3)
            ev[k].append(ev_ms)
    med = lambda v: float(np.median(v))
    ms = med(wall["plain"])
    res = {"workload": "combsub B=%d/GPU x %.0f s on %d GPU(s): %d utterances, n_mag %d/%d/%d (BASELINE cfg 4)"
                       % (B4, a.seconds, world, B4 * world, n, n, n),
           "batch_per_gpu": B4, "n_gpus": world, "steps": per * rounds, "rounds": rounds, "steps_per_round": per, "warmup": 3,
           "method": "interleaved rounds (plain, gather, gather_async, plain, ..) of %d steps after clock ramp-up; medians over rounds" % per,
           "ms_per_step": ms, "ms_per_step_events": med(ev["plain"]), "value": B4 * world * T / (ms * 1e-3), "unit": "samples/s",
           "rounds_ms": [round(v, 4) for v in wall["plain"]], "rccl_ranks": comm.get("rccl_ranks")}
    if have_pg:
        alone, y = [], step()
        fence()
        for _ in range(3):                                               # the gather alone, between fences
            t1 = time.perf_counter()
            full = sharding.gather_utterances(y, B4 * world, dst=0)
            fence()
            alone.append((time.perf_counter() - t1) * 1e3)
        (g_alone,) = reduce_max(med(alone))
        if rank == 0:
            assert full.shape == (B4 * world, T) and torch.isfinite(full[::8, ::4096]).all()
            res["gather_checked_rows"] = int(full.shape[0])
        msg = med(wall["gather"])
        res.update({"ms_per_step_with_gather": msg, "ms_per_step_with_gather_events": med(ev["gather"]),
                    "value_with_gather": B4 * world * T / (msg * 1e-3),
                    "rounds_ms_with_gather": [round(v, 4) for v in wall["gather"]],
                    "gather_ms": g_alone,
                    "gather_overhead_ms": med([g - p for g, p in zip(wall["gather"], wall["plain"])]),
                    "ms_per_step_with_gather_async": med(wall["gather_async"]),
                    "gather_async_overhead_ms": med([g - p for g, p in zip(wall["gather_async"], wall["plain"])]),
                    "gather_bytes_per_rank": 4.0 * B4 * T,
                    "ms_per_step_with_gather_in_place": med(wall["gather_in_place"]) if "gather_in_place" in wall else None,
                    "gather_in_place_overhead_ms": med([g - p for g, p in zip(wall["gather_in_place"], wall["plain"])])
                    if "gather_in_place" in wall else None,
                    "gather_in_place": "the receiving rank keeps the result tensor from step to step and its synthesis writes its "
                                       "own shard straight into its slice (signal_out= / out=): no copy of the local shard -- at one "
                                       "rank nothing moves" + ("" if "gather_in_place" in wall else
                                                                "; NOT timed: %s" % comm.get("cfg4_in_place_error")),
                    "gather": "torch.distributed.gather (%s) of every step's [%d, T] waveforms into slices of the "
                              "result on rank 0, waited for before the next step starts; *_async: step k's gather issued async, "
                              "running under step k + 1's synthesis and waited for one step later (the last one inside the timed "
                              "region); gather_ms = one gather alone between fences "
                              "(median of 3), gather_overhead_ms = median of the rounds' paired differences (what a step pays)"
                              % ("nccl = RCCL" if RT.backend == "nccl" else RT.backend, B4)})
    else:
        res["gather"] = "not timed: no process group (%s)" % comm.get("cfg4_communicator_error", "--no-cfg4-gather")
    return res


def parity_gate(kind, f0, ctrls, noise, out, rows=None):
    """no timing is printed for an output that is not the reference's.  Utterances ``rows`` (default: the
    first and the last -- the last one is where a wrong sub-batch offset would land) of the TIMED step's output against the
    oracle's restatement of the reference on the same f0 / controls / noise: <= 1e-4 RMS absolute (the north star's bar) and
    <= 1e-5 relative, else SystemExit.  Returns the record that goes into the line as ``parity_vs_oracle``."""
    from oracle import ddsp_oracle as O
    B = f0.shape[0]
    rows = sorted(set(rows if rows is not None else (0, B - 1)))
    t0 = time.perf_counter()
    fn = O.combsub_dsp if kind == "combsub" else O.sins_dsp
    worst_abs = worst_rel = 0.0
    for b in rows:
        args = [t[b:b + 1].detach().cpu().numpy() for t in (f0, ctrls[0], ctrls[1], ctrls[2], noise)]
        ref = fn(np.ascontiguousarray(args[0]), *(np.ascontiguousarray(c) for c in args[1:4]),
                 np.ascontiguousarray(args[4]), SR, HOP)["signal"].astype(np.float64)
        got = out[b:b + 1].detach().cpu().numpy().astype(np.float64)
        e = float(np.sqrt(np.mean((got - ref) ** 2)))
        r = e / max(float(np.sqrt(np.mean(ref ** 2))), 1e-30)
        worst_abs, worst_rel = max(worst_abs, e), max(worst_rel, r)
    rec = {"rows": rows, "rms_abs": worst_abs, "rms_rel": worst_rel, "bar_abs": 1e-4, "bar_rel": 1e-5,
           "oracle": "oracle/ddsp_oracle.py %s_dsp (numpy restatement of the reference, pinned to its fixtures)" % kind,
           "seconds": time.perf_counter() - t0}
    if not (worst_abs <= 1e-4 and worst_rel <= 1e-5) or not np.isfinite(worst_abs):
        raise SystemExit("bench.py: the timed output is NOT the reference's: %s" % json.dumps(rec))
    return rec


def cpu_baseline(kind, F, sizes, budget_s=12.0):
    """Oracle timed on the host: one 10 s utterance per worker process, rounds until ~budget."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 64))
    ctx = mp.get_context("fork")
    done, t0 = 0, time.perf_counter()
    with ctx.Pool(cores, initializer=_cpu_worker_init) as pool:
        pool.map(_cpu_worker, [(1000 + i, 8, kind, sizes) for i in range(cores)])     # warm the workers
        t0 = time.perf_counter()
        rounds = 0
        while True:                                 # whole rounds (one utterance per core) until ~budget_s of wall time
            pool.map(_cpu_worker, [(rounds * cores + i, F, kind, sizes) for i in range(cores)])
            rounds += 1
            done += cores
            el = time.perf_counter() - t0
            if el + el / rounds > budget_s or rounds >= 64:
                break
        wall = time.perf_counter() - t0
    return {"value": done * F * HOP / wall, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d utterances of %d frames (%.1f s audio each), numpy oracle, %d worker processes, %.1f s wall"
                      % (done, F, F * HOP / SR, cores, wall)}


def _reference_root():
    """a reference checkout that travelled with this run (tools/with_reference.sh sets DDSP_REFERENCE_PATH), or None.
    /root/reference itself is never read here: it does not exist on the GPU box."""
    r = os.environ.get("DDSP_REFERENCE_PATH")
    return r if r and os.path.isdir(os.path.join(r, "ddsp")) else None


def _import_reference_vocoder(root):
    from unittest.mock import MagicMock
    if root not in sys.path:
        sys.path.insert(0, root)
    for name in ["transformers", "pyworld", "parselmouth", "torchcrepe", "resampy", "fairseq", "torchaudio", "torchaudio.transforms",
                 "gin", "local_attention", "librosa", "librosa.sequence", "librosa.util", "librosa.filters", "librosa.core",
                 "soundfile"]:
        sys.modules.setdefault(name, MagicMock())      # third-party imports of the reference that this image lacks
    import ddsp.vocoder as rvoc
    return rvoc


def cpu_baseline_reference(kind, F, n, device, budget_s=25.0):
    """the UNMODIFIED reference module (ddsp/vocoder.py Sins / CombSub, random-init Unit2Control, eval,
    no_grad, float32, infer=True) on this host's cores, same process and run as the GPU number -- full forward and DSP only
    (forward minus the separately timed unit2ctrl) -- and the parity gate of 3.7 on the same inputs: the HIP path on the
    controls the reference's own Unit2Control produced, against the reference's waveform (must be <= 1e-4 RMS)."""
    from unittest import mock
    root = _reference_root()
    rvoc = _import_reference_vocoder(root)
    from ddsp_svc_amd import synth
    name = "CombSub" if kind == "combsub" else "Sins"
    cls = getattr(rvoc, "_reference_" + name, getattr(rvoc, name))
    cores = os.cpu_count() or 1
    torch.manual_seed(1234)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):         # the reference announces its models on stdout; the JSON line owns that
        ref = cls(SR, HOP, n, n, n, n_unit=768, n_spk=1).eval()
    T = F * HOP
    g = torch.Generator().manual_seed(1234)
    t_u2c = [0.0]
    captured = {}

    def pre(mod, args):
        t_u2c.append(time.perf_counter())

    def post(mod, args, out):
        t_u2c[0] += time.perf_counter() - t_u2c.pop()
        captured["ctrls"] = out[0]
    ref.unit2ctrl.register_forward_pre_hook(pre)
    ref.unit2ctrl.register_forward_hook(post)

    def run(B):
        units = torch.randn(B, F, 768, generator=g)
        vol = torch.rand(B, F, 1, generator=g) * 0.1
        f0 = torch.from_numpy(synthetic_f0(B, F, 1234))
        u = torch.rand(B, T, generator=g)
        t_u2c[0] = 0.0
        t0 = time.perf_counter()
        with torch.no_grad(), mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)):
            sig, _, _ = ref(units, f0, vol, infer=True)
        return time.perf_counter() - t0, t_u2c[0], sig, f0, u
    # BASELINE.md 3.2 asks for torch.set_num_threads(os.cpu_count()); on a 256-thread host that is the WORST setting for these
    # op sizes (measured: 35 s per 10 s utterance, the intra-op pool oversubscribed), so the pool size is chosen like the
    # op-chain leg's: the fastest of a few sizes on a two-utterance run, and the choice is reported (`cores`)
    B = 2
    best = None
    for th in sorted({min(cores, 8), min(cores, 32), min(cores, 64)}):      # (all 256 hardware threads: 35 s per utterance, not tried again)
        torch.set_num_threads(th)
        run(B)
        w = run(B)[0]
        if best is None or w < best[0]:
            best = (w, th)
    threads = best[1]
    torch.set_num_threads(threads)
    wall, u2c, sig, f0, u = run(B)                      # the parity gate's inputs
    c = [v.to(device) for v in captured["ctrls"].values()]
    st = synth.phase(f0.to(device), SR, HOP)
    tail = synth.combsub_synth if kind == "combsub" else synth.sins_synth
    ours = tail(f0.to(device), st, c[0], c[1], c[2], u.to(device), SR, HOP, noise_is_u01=True, want_components=False)[0].cpu()
    d = (ours - sig).double()
    parity = {"rms_error": float(d.pow(2).mean().sqrt()), "max_abs_error": float(d.abs().max()),
              "rms_reference": float(sig.double().pow(2).mean().sqrt()), "utterances": B, "bar": 1e-4}
    parity["relative_rms"] = parity["rms_error"] / max(parity["rms_reference"], 1e-30)
    if not parity["rms_error"] <= 1e-4:
        raise SystemExit("parity gate failed against the reference module: %r" % parity)
    # size the timed batch so that three runs fit the budget
    B = max(1, min(32, int(B * (budget_s / 4.0) / max(wall, 1e-3))))
    walls, dsps = [], []
    for _ in range(3):
        w, c_, _, _, _ = run(B)
        walls.append(w)
        dsps.append(w - c_)
    w_med, d_med = sorted(walls)[1], sorted(dsps)[1]
    cpu = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": B * T / d_med, "unit": "samples/s", "cores": threads, "kind": "reference",
            "value_full_forward": B * T / w_med,
            "sample": "the unmodified reference %s(44100, 512, %d, %d, %d, n_unit 768) on CPU tensors, %d utterances of %.1f s "
                      "per run, 1 warm-up + 3 runs, median: full forward %.2f s, unit2ctrl %.2f s, DSP only %.2f s; value = DSP only"
                      % (name, n, n, n, B, T / SR, w_med, w_med - d_med, d_med),
            "torch_threads": torch.get_num_threads(), "host_logical_cpus": cores, "host_cpu": cpu, "torch": torch.__version__,
            "parity_vs_reference": parity}


def _cpu_mel_worker(args):
    seed, T, ks = args
    import numpy as _np
    from oracle import ddsp_oracle as O
    y = O.synth_gauss(1, T, seed=seed) * _np.float32(0.1)
    return float(O.get_mel(y, O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000), keyshift=ks).max())


def step_traffic(model):
    """PMC HBM bytes of ONE whole step of ``model`` from the newest committed profiles/*_hbm_traffic.json that carries a
    ``__step__`` entry (tools/gpu_traffic.sh over ``bench.py --only-steps``: every kernel's bytes x its launches / steps)."""
    pdir = os.path.join(ROOT, "profiles")
    import re
    best = None
    for name in sorted(os.listdir(pdir), key=lambda n: [int(x) for x in re.findall(r"\d+", n)]) if os.path.isdir(pdir) else []:
        if name.endswith("_hbm_traffic.json"):
            try:
                d = json.load(open(os.path.join(pdir, name))).get("__step__")
            except Exception:
                continue
            if d and d.get("model") == model:
                best = dict(d, source="profiles/" + name)
    return best


def cpu_baseline_aten_chain(kind, F, sizes, budget_s=14.0):
    """The reference's op chain (F.interpolate, float64 cumsum, rfft / irfft at 2 hop + N - 1 points, fold) walked with
    torch CPU operators on this host (oracle/aten_chain.py, pinned to the reference's fixtures): what the reference's DSP
    tail costs here, without the reference checkout.  BASELINE.md section 3 prescribes ``torch.set_num_threads(os.cpu_count())``;
    on a many-core host that oversubscribes ATen's intra-op pool (measured: 256 threads are slower than 32), so the
    prescribed setting and a few smaller pools are each timed on a bounded batch and the best one is reported (all
    listed in ``thread_sweep``)."""
    from oracle import aten_chain as A
    cores = os.cpu_count() or 1
    prev = torch.get_num_threads()
    batch = 8 if kind == "combsub" else 4

    def run(B, seed):
        g = torch.Generator().manual_seed(seed)
        f0 = torch.from_numpy(synthetic_f0(B, F, seed))
        c = [torch.randn(B, F, n, generator=g) for n in sizes]
        nz = torch.rand(B, F * HOP, generator=g) * 2 - 1
        fn = A.sins_tail if kind == "sins" else A.combsub_tail
        t0 = time.perf_counter()
        with torch.no_grad():
            out = fn(f0, c[0], c[1], c[2], nz, SR, HOP, True)[0]
        dt = time.perf_counter() - t0
        assert torch.isfinite(out).all()
        return dt
    sweep = {}
    t_start = time.perf_counter()
    try:
        for threads in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16)}):      # the small pools first: they are the fast ones
            if time.perf_counter() - t_start > budget_s and sweep:
                break
            torch.set_num_threads(threads)
            run(1, 1)
            sweep[threads] = batch * F * HOP / min(run(batch, 100 + i) for i in range(2))
    finally:
        torch.set_num_threads(prev)
    best = max(sweep, key=sweep.get)
    return {"value": sweep[best], "unit": "samples/s", "cores": cores, "kind": "aten-chain", "threads": best,
            "thread_sweep": {str(k): v for k, v in sweep.items()},
            "prescribed_threads_timed": cores in sweep,
            "sample": "best of 2 runs of B=%d x %d frames (%.1f s audio each) through the reference's op chain with torch %s "
                      "CPU operators, per intra-op pool size; %.1f s wall in total" %
                      (batch, F, F * HOP / SR, torch.__version__, time.perf_counter() - t_start)}


def aten_chain_on_gpu(kind, f0, ctrls, noise, ours, device, steps=5):
    """Second half of the ``cpu_baseline_aten_chain`` leg: the SAME op chain with its tensors on the MI355X -- what a user gets
    from running the reference's DSP tail as it is under PyTorch-ROCm (rocFFT at 2 hop + N - 1 = 1533 points, fold, ...),
    on the full workload of the headline, beside the HIP path's step and compared with its output."""
    from oracle import aten_chain as A
    fn = A.sins_tail if kind == "sins" else A.combsub_tail
    B, F = f0.shape[0], f0.shape[1]
    f0c = f0.reshape(B, F, 1)

    def step():
        with torch.no_grad():
            return fn(f0c, ctrls[0], ctrls[1], ctrls[2], noise, SR, HOP, True)[0]
    t_first = time.perf_counter()
    out = step()                                                    # rocFFT plans, allocator growth
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    err = float((out - ours).pow(2).mean().sqrt() / ours.pow(2).mean().sqrt())
    peak = torch.cuda.max_memory_allocated(device) / 2 ** 30
    return {"ms_per_step": ms, "value": B * F * HOP / (ms * 1e-3), "unit": "samples/s", "steps": steps,
            "first_call_s": t_first, "rel_rms_vs_hip_path": err, "peak_memory_gib": peak,
            "what": "oracle/aten_chain.py (the reference's op sequence) with torch %s ROCm operators on the same GPU, same "
                    "inputs, B=%d x %d frames" % (torch.__version__, B, F)}


def module_mode(kind, B, F, n, device, steps, warmup):
    """Control mode (i) of SURVEY.md 8-d: the drop-in module end to end -- HOT-1, a Unit2Control producing the controls on
    the GPU, torch.rand noise, HOT-2 -- with a stand-in of the reference's Unit2Control shape (tools/standins.py: the
    reference's own module cannot travel to the GPU box)."""
    from ddsp_svc_amd import vocoder as V
    from tools.standins import StandInUnit2Control
    torch.manual_seed(0)
    if kind == "sins":
        m = V.Sins(SR, HOP, n, n, n, n_unit=768, n_spk=1, unit2ctrl_factory=StandInUnit2Control)
    else:
        m = V.CombSub(SR, HOP, n, n, n, n_unit=768, n_spk=1, unit2ctrl_factory=StandInUnit2Control)
    m = m.to(device).eval()
    m.return_components = False
    g = torch.Generator().manual_seed(5)
    units = torch.randn(B, F, 768, generator=g).to(device)
    vol = (torch.rand(B, F, 1, generator=g) * 0.1).to(device)
    f0 = torch.from_numpy(synthetic_f0(B, F, 1234)).to(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.no_grad():
        for _ in range(warmup):
            out = m(units, f0, vol)[0]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(steps):
            out = m(units, f0, vol)[0]
        ev[1].record()
        ph = torch.zeros(B, F, 1, device=device)
        ev[2].record()
        for _ in range(steps):
            m.unit2ctrl(units, f0, ph, vol)
        ev[3].record()
        torch.cuda.synchronize()
        # the same with the noise drawn inside the noise filter instead of a torch.rand tensor (opt-in, in_kernel_noise_seed)
        m.in_kernel_noise_seed = 1234
        for _ in range(warmup):
            out2 = m(units, f0, vol)[0]
        torch.cuda.synchronize()
        e4, e5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e4.record()
        for _ in range(steps):
            out2 = m(units, f0, vol)[0]
        e5.record()
        torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(out2).all()
    ms = ev[0].elapsed_time(ev[1]) / steps
    return {"value": B * F * HOP / (ms * 1e-3), "unit": "samples/s", "ms_per_step": ms,
            "unit2ctrl_ms": ev[2].elapsed_time(ev[3]) / steps,
            "ms_per_step_in_kernel_noise": e4.elapsed_time(e5) / steps,
            "note": "drop-in %s module forward (HOT-1 + stand-in Unit2Control of %.1f M parameters + torch.rand + HOT-2), "
                    "signal only; the stand-in is not the reference's network" %
                    (type(m).__name__, sum(p.numel() for p in m.unit2ctrl.parameters()) / 1e6)}


def launch_ranks(n):
    """``--gpus N`` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    have = RT.device_count()
    if have < n:
        raise SystemExit("bench.py --gpus %d needs %d devices, this host shows %d" % (n, n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def setup_ranks(a):
    """(rank, world, device, info): reads the launcher's environment, refuses a world size that is not ``--gpus``, opens
    the RCCL communicator and proves it with one all-reduce."""
    import torch.distributed as dist
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(a.gpus)                              # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: refusing to report a different GPU count"
                         % (a.gpus, world))
    if not RT.available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if RT.device_count() <= local_rank:
        raise SystemExit("bench.py --gpus %d needs %d devices, this host shows %d" % (a.gpus, a.gpus, RT.device_count()))
    device = RT.device(local_rank)
    info = {}
    if world > 1 or a.gather or a.cfg4:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        RT.init_process_group(rank, world, device)
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        info["rccl_ranks"] = int(probe.item())
        if info["rccl_ranks"] != world:
            raise SystemExit("%s all-reduce saw %d ranks, expected %d" % (RT.backend, info["rccl_ranks"], world))
    return rank, world, device, info


def finish_ranks():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


_PENDING = []


def emit(res):
    """queue rank 0's JSON line; it is printed by flush_emit() after the communicator is gone and C stdio is flushed, so
    that nothing a library prints (RCCL's version banner goes through buffered C stdout) can land after it"""
    _PENDING.append(json.dumps(res))


def flush_emit():
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    for line in _PENDING:
        print(line, flush=True)
    _PENDING.clear()


def bench_sinesrc(a, rank, world, device):
    """harmonic source of NSF-HiFiGAN: SineGen + merge for B x 10 s, 9 harmonics.  What the caller PAYS per
    call is timed, i.e. with the standard-normal draw of models.py:168 inside the step, in both forms: (a) the reference's way,
    ``torch.randn(B, T, 9)`` into HBM and the kernel reading it back (36 B per sample written, 36 read, 4 written); (b) opt-in,
    the draw inside the kernel (Philox + Box-Muller: 4 B per sample written, nothing of size [B, T, 9] allocated).  ``value`` is
    (a), the default of the drop-in; (b) is ``in_kernel_noise``; the kernel alone on a resident draw (round 4's line) ``kernel_only``."""
    import torch.distributed as dist
    from ddsp_svc_amd import nsf_source as S
    B = a.batch_per_gpu
    F = int(a.seconds * SR) // HOP + 1
    T = F * HOP
    g = torch.Generator(device="cpu").manual_seed(777 + rank)
    f0 = torch.from_numpy(synthetic_f0(B, F, 55 + rank)[..., 0]).to(device)
    w = (torch.randn(9, generator=g) * 0.3).to(device)
    b = torch.zeros(1, device=device)
    ri = torch.rand(9, generator=g).to(device)
    ri[0] = 0
    calls = [0]

    def step_randn():
        return S.sine_source(f0, HOP, SR, w, b, ri, torch.randn(B, T, 9, device=device))

    def step_drawn():
        calls[0] += 1
        return S.sine_source(f0, HOP, SR, w, b, ri, None, noise_seed=1234 + rank, noise_offset=calls[0])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        prewarm(step, a.prewarm_seconds)
        torch.cuda.reset_peak_memory_stats(device)
        base = torch.cuda.memory_allocated(device)
        el, ev_ms, out = time_steps(step, a.steps, a.warmup, fence)
        peak = torch.cuda.max_memory_allocated(device) - base
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        assert torch.isfinite(out).all()
        return el, ev_ms, peak, out
    el_a, ev_a, peak_a, out_a = timed(step_randn)
    el_b, ev_b, peak_b, out_b = timed(step_drawn)
    noise = torch.randn(B, T, 9, device=device)
    el_k, ev_k, _, _ = timed(lambda: S.sine_source(f0, HOP, SR, w, b, ri, noise))
    # the drawn variant is the same kernel fed its own numbers (bit for bit), and those numbers are standard normal
    z = S.normal_noise(min(B, 2), T, 9, 1234 + rank, calls[0], device)
    same = torch.equal(out_b[:z.shape[0]], S.sine_source(f0[:z.shape[0]], HOP, SR, w, b, ri, z))
    moments = (float(z.mean()), float(z.var()))
    del noise, z
    if rank != 0:
        return
    if not same or abs(moments[0]) > 1e-2 or abs(moments[1] - 1.0) > 1e-2:
        raise SystemExit("bench.py: the in-kernel draw is not what the kernel fed the written-out draw gives (%s, moments %s)" % (same, moments))
    ms_a, ms_b, ms_k = el_a / a.steps * 1e3, el_b / a.steps * 1e3, el_k / a.steps * 1e3
    alg_a, alg_b = (36.0 + 36.0 + 4.0) * B * T, 4.0 * B * T + 4.0 * B * F
    emit(({
        "metric": "audio samples/sec, NSF-HiFiGAN harmonic source 44.1kHz upp512 9 harmonics",
        "value": B * world * T * a.steps / el_a, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "prewarm_s": a.prewarm_seconds, "ms_per_step": ms_a, "ms_per_step_events": ev_a, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SourceModuleHnNSF.forward for B=%d/GPU x %.0f s (T=%d), 9 harmonics, torch.randn draw INSIDE the "
                               "timed step (models.py:168)" % (B, a.seconds, T), "batch_per_gpu": B, "samples_per_utterance": T,
                   "parallelism": "utterance-shard x%d" % world},
        "peak_bytes_per_call": peak_a,
        "roofline": {"kernel": "randn + k_sinegen<9,false>", "bound": "hbm", "achieved": alg_a / (ms_a * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": alg_a / (ms_a * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_a, "avg_ms": ms_a, "launches_per_step": 2,
                     "note": "bytes of THIS form (the draw written, read back, the sample written); the path's own algorithmic "
                             "bytes are the 4 B per sample of in_kernel_noise"},
        "in_kernel_noise": {"ms_per_step": ms_b, "ms_per_step_events": ev_b, "value": B * world * T * a.steps / el_b,
                            "peak_bytes_per_call": peak_b, "same_kernel_fed_the_written_out_draw": same,
                            "draw_moments": {"mean": moments[0], "var": moments[1]},
                            "roofline": {"kernel": "k_sinegen<9,true>", "bound": "hbm", "achieved": alg_b / (ms_b * 1e-3) / 1e9,
                                         "peak": 8000.0, "unit": "GB/s", "frac": alg_b / (ms_b * 1e-3) / 1e9 / 8000.0,
                                         "algorithmic_bytes_per_launch": alg_b,
                                         "note": "not HBM-bound any more: 3 Philox4x32-10 + 5 Box-Muller pairs + 9 sines per sample"}},
        "kernel_only": {"ms_per_step": ms_k, "note": "k_sinegen<9,false> on a resident draw (the draw untimed: round 4's line)"}}))


def bench_rssloss(a, rank, world, device):
    """training loss: RSSLoss forward + backward w.r.t. the prediction for B x 10 s, the four
    transform sizes fixed (a draw of torch.randint(256, 2048, (4,))); the eager composition of loss.py:22-31 on the
    same device is timed beside it"""
    import torch.distributed as dist
    from ddsp_svc_amd import loss as L
    B = a.batch_per_gpu
    F = int(a.seconds * SR) // HOP + 1
    T = F * HOP
    g = torch.Generator(device="cpu").manual_seed(99 + rank)
    xt = (torch.randn(B, T, generator=g) * 0.1).to(device)
    xp = (xt * 0.9 + 0.02 * torch.randn(B, T, generator=g).to(device)).requires_grad_(True)
    sizes = [1153, 397, 2011, 768]
    fs = [L.SSSLoss(n).to(device) for n in sizes]
    rss = L.RSSLoss(256, 2048, len(sizes), device=device)
    drawn = torch.tensor(sizes)
    real_randint = torch.randint

    def step():
        torch.randint = lambda *a_, **k_: drawn                      # loss.py:47's draw, pinned to the four sizes above
        try:
            loss = rss(xp, xt)
        finally:
            torch.randint = real_randint
        grad, = torch.autograd.grad(loss, xp)
        return loss, grad

    def eager():
        value = 0.
        for f in fs:
            n, w = f.n_fft, f.spec.window
            sp = lambda x: torch.stft(x, n, hop_length=n, win_length=n, window=w, center=False,
                                      return_complex=True).abs() / w.pow(2).sum().sqrt() + 1e-7
            St, Sp = sp(xt), sp(xp)
            value = value + torch.mean(torch.linalg.norm(St - Sp, dim=(1, 2)) / torch.linalg.norm(St + Sp, dim=(1, 2))) \
                + torch.nn.functional.l1_loss(St.log(), Sp.log())
        loss = value / len(fs)
        grad, = torch.autograd.grad(loss, xp)
        return loss, grad

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        prewarm(fn, a.prewarm_seconds)
        for _ in range(a.warmup):
            out = fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        fence()
        return time.perf_counter() - t0, out
    elapsed, (loss, grad) = timed(step, a.steps)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(loss) and torch.isfinite(grad).all()
    if rank != 0:
        return
    e_elapsed, (e_loss, e_grad) = timed(eager, max(2, a.steps // 4))
    e_ms = e_elapsed / max(2, a.steps // 4) * 1e3
    # the kernels alone: forward (no graph) by events; the backward kernels are the rest of a fused step's GPU time
    in_kernel = os.environ.get("DDSP_HIP_LOSS_TORCH_STFT", "").strip() in ("", "0")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.randint = lambda *a_, **k_: drawn
    try:
        torch.cuda.synchronize()
        ev[0].record()
        with torch.no_grad():
            for _ in range(10):
                rss(xp, xt)
        ev[1].record()
        for _ in range(10):
            torch.autograd.grad(rss(xp, xt), xp)
        ev[2].record()
        torch.cuda.synchronize()
    finally:
        torch.randint = real_randint
    f_ms = ev[0].elapsed_time(ev[1]) / 10
    fb_ms = ev[1].elapsed_time(ev[2]) / 10
    # algorithmic bytes of one step: forward 8 B per sample pair read + 16 B per bin pair written; backward 16 B per bin
    # pair read + 4 B per sample written (+ 4 B read where a scale adds into the gradient)
    bins = sum((n // 2 + 1) * (T // n) for n in sizes) * B
    alg = 8.0 * B * T * len(sizes) + 16.0 * bins + 16.0 * bins + 4.0 * B * T * (2 * len(sizes) - 1)
    # arithmetic: per frame (pair) two complex transforms of N points, 5 N log2 N flops each
    def plan(n):
        return 1024 if 2 * n - 1 <= 1024 else (2048 if 2 * n - 1 <= 2048 else 4096)
    flops = sum((T // n) * B * 1.5 * 2 * 5.0 * plan(n) * math.log2(plan(n)) for n in sizes)
    ms = elapsed / a.steps * 1e3
    emit(({
        "metric": "audio samples/sec, RSSLoss forward+backward 44.1kHz 4 scales",
        "value": B * world * T * a.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "prewarm_s": a.prewarm_seconds, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "RSSLoss(x_pred, x_true) + d/dx_pred for B=%d/GPU x %.0f s (T=%d), transform sizes %s "
                               "(hop = size), %s"
                               % (B, a.seconds, T, sizes, "STFT inside the loss kernels (chirp-z, csrc/loss_czt.hip)" if in_kernel
                                  else "STFT by torch.stft (rocFFT), everything behind it fused"),
                   "batch_per_gpu": B, "samples_per_utterance": T, "parallelism": "utterance-shard x%d" % world},
        "roofline": {"kernel": "k_sss_czt + k_sss_czt_bwd (4 scales each)" if in_kernel else "torch.stft + k_sss_partial / k_sss_grad",
                     "bound": "hbm", "achieved": alg / (fb_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (fb_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": alg, "avg_ms": fb_ms, "forward_only_ms": f_ms, "launches_per_step": 1,
                     "note": "the kernels are bound by float32 vector arithmetic, not by either roof of the contract: "
                             "%.1f GFLOP of transforms per step = %.1f TFLOP/s (vector float32 peak 157)"
                             % (flops / 1e9, flops / (fb_ms * 1e-3) / 1e12)},
        "eager_composition": {"ms_per_step": e_ms, "speedup": e_ms / ms,
                              "loss_rel_diff": abs(float(loss) - float(e_loss)) / float(e_loss),
                              "grad_rel_rms": float((grad - e_grad).pow(2).mean().sqrt() / e_grad.pow(2).mean().sqrt()),
                              "grad_note": "the loss's gradient is discontinuous where S_true = S_pred (sign of the log "
                                           "difference); bins within float32 rounding of that flip between any two "
                                           "float32 transforms"}}))


def bench_mel(a, rank, world, device):
    """waveform -> log-mel front-end of the cascade: one k_mel launch over B x 10 s of audio; with
    --keyshift k (main_diff.py:359's formant shift): one k_mel_czt launch, a transform of round(2048 2^(k/12)) points"""
    import multiprocessing as mp
    import torch.distributed as dist
    from ddsp_svc_amd import mel as M
    B = a.batch_per_gpu
    F = int(a.seconds * SR) // HOP + 1
    T = F * HOP
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    y = (torch.randn(B, T, generator=g) * 0.1).to(device)
    stft = M.STFT(44100, 128, 2048, 2048, 512, 40, 16000)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    ks = a.keyshift
    prewarm(lambda: stft.get_mel(y, keyshift=ks), a.prewarm_seconds)
    for _ in range(a.warmup):
        out = stft.get_mel(y, keyshift=ks)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = stft.get_mel(y, keyshift=ks)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        stft.get_mel(y, keyshift=ks)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 20
    if rank != 0:
        return
    alg = (4.0 + 4.0 * 128 / HOP) * B * T                       # waveform in, [B,F,128] log-mel out
    ms = elapsed / a.steps * 1e3
    n_new = M._shifted_sizes(2048, 2048, 512, ks, 1)[0]
    res = {"metric": "audio samples/sec, log-mel front-end 44.1kHz n_fft2048 hop512 128 mels" +
                     ("" if ks == 0 else ", keyshift %g (a %d-point transform)" % (ks, n_new)),
           "value": B * world * T * a.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "prewarm_s": a.prewarm_seconds, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "STFT.get_mel of B=%d/GPU x %.0f s waveforms (T=%d) -> [B,128,%d] log-mel"
                                  % (B, a.seconds, T, F), "batch_per_gpu": B, "samples_per_utterance": T,
                      "parallelism": "utterance-shard x%d" % world},
           "roofline": {"kernel": "k_mel" if ks == 0 else "k_mel_czt", "bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
                        "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                        "algorithmic_bytes_per_launch": alg, "avg_ms": k_ms, "launches_per_step": 1}}
    if world == 1 and not a.no_cpu_baseline:
        cores = max(1, min(os.cpu_count() or 1, 64))
        with mp.get_context("fork").Pool(cores, initializer=_cpu_worker_init) as pool:
            pool.map(_cpu_mel_worker, [(i, 4096, ks) for i in range(cores)])
            t1 = time.perf_counter()
            rounds = 0
            while time.perf_counter() - t1 < 8.0 and rounds < 64:
                pool.map(_cpu_mel_worker, [(rounds * cores + i, T, ks) for i in range(cores)])
                rounds += 1
            wall = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": rounds * cores * T / wall, "unit": "samples/s", "cores": cores, "kind": "port",
                               "sample": "%d waveforms of %.1f s, numpy oracle get_mel, %d worker processes, %.1f s wall"
                                         % (rounds * cores, T / SR, cores, wall)}
    emit((res))


def report_fast(a, rank, world, B, F, T, sizes, elapsed, gather_ms, f0, ctrls, noise, window, win):
    """JSON line for CombSubFast / CombSubSuperFast: the dominant kernel is the fused
    short-time spectral filter k_stft_filter, timed alone with events on the launch stream."""
    from ddsp_svc_amd import synth
    exc = torch.randn(B, T, device=f0.device)
    super_ = a.model == "combsubsuperfast"

    def once():
        return synth.stft_filter(exc, noise, ctrls[0], ctrls[1], ctrls[2], ctrls[3] if super_ else None, window, HOP,
                                 pad_reflect=super_, normalize=super_)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / reps
    if rank != 0:
        return
    sigma_c = sum(sizes)
    # the kernel reads exciter + noise + controls and writes the signal; the step regenerates the exciter from f0
    k_bytes = (12.0 + 4.0 * sigma_c / HOP) * B * T
    alg_bytes = (8.0 + 4.0 * (sigma_c + 1) / HOP) * B * T
    kname = "k_stft_filter<%d>" % (4 if win == 2048 else 2)
    traffic = hbm_traffic("k_stft_filter") if (B, F) == (32, 862) and super_ else None
    pairs = (F + 2) // 2
    fft_flops = B * pairs * (3 * 5.0 * win * np.log2(win) + win * 30.0)
    total = B * world * T * a.steps
    value = total / elapsed
    ms = elapsed / a.steps * 1e3
    res = {
        "metric": "audio samples/sec, %s 44.1kHz win%d hop512" % ("CombSubSuperFast" if super_ else "CombSubFast", win),
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "prewarm_s": a.prewarm_seconds,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s B=%d/GPU x %.0f s utterances (F=%d, T=%d), %d x %d control bins, sr 44100, hop 512, "
                               "DSP path (exciter phase + exciter + short-time spectral filter) from resident f0 / raw "
                               "controls ~N(0,1) / noise" % (a.model, B, a.seconds, F, T, len(sizes), sizes[0]),
                   "batch_per_gpu": B, "frames": F, "samples_per_utterance": T, "parallelism": "utterance-shard x%d" % world},
        "roofline": {"kernel": kname, "bound": "hbm", "achieved": k_bytes / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": k_bytes / (k_ms * 1e-3) / 1e9 / 8000.0,
                     "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
                     "algorithmic_bytes_per_launch": k_bytes, "avg_ms": k_ms, "launches_per_step": 1},
        "roofline_compute": {"kernel": kname, "bound": "valu", "unit": "TFLOP/s", "peak": 157.3,
                             "achieved": fft_flops / (k_ms * 1e-3) / 1e12, "frac": fft_flops / (k_ms * 1e-3) / 1e12 / 157.3,
                             "executed_flops_per_launch": fft_flops},
        "roofline_step_hbm": {"bound": "hbm", "algorithmic_bytes_per_step": alg_bytes,
                              "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": alg_bytes / (ms * 1e-3) / 1e9 / 8000.0},
    }
    if gather_ms is not None:
        res["gather_ms"] = gather_ms
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.model, F, sizes)
        res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
    emit((res))


def bench_cascade_seam(a, rank, world, device):
    """BASELINE cfg 5's seam on one MI355X (main_diff.py:356-359,378): drop-in CombSubSuperFast -> log-mel -> denoiser ->
    NSF harmonic source + generator body, B utterances of 10 s on one stream.  The neural parts are stand-ins
    (tools/standins.py: the reference's networks and checkpoints cannot travel to the GPU box); the DSP parts are the
    product's kernels.  Reported: the whole chain, and the part of it the HIP kernels account for."""
    import torch.distributed as dist
    from tools.standins import CascadeSeam
    B = a.batch_per_gpu
    F = int(a.seconds * SR) // HOP + 1
    T = F * HOP
    torch.manual_seed(0)
    seam = CascadeSeam(SR, HOP).to(device).eval()
    g = torch.Generator().manual_seed(31 + rank)
    units = torch.randn(B, F, 768, generator=g).to(device)
    vol = (torch.rand(B, F, 1, generator=g) * 0.1).to(device)
    f0 = torch.from_numpy(synthetic_f0(B, F, 1234 + rank)).to(device)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    steps, warm = max(3, a.steps // 10), max(2, a.warmup // 5)
    for _ in range(warm):
        out = seam(units, f0, vol)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = seam(units, f0, vol)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    wav, ddsp_wav, ddsp_mel = out
    assert wav.shape == (B, T) and ddsp_mel.shape == (B, F, 128)
    assert torch.isfinite(wav).all() and torch.isfinite(ddsp_wav).all() and torch.isfinite(ddsp_mel).all()
    # the DSP pieces alone (the product's share of the chain), each on resident inputs
    from ddsp_svc_amd import synth
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.no_grad():
        st = synth.fast_source(f0, SR, HOP)
        ctrls, _ = seam.ddsp.unit2ctrl(units, f0, st.phase_frames, vol)
        gauss = torch.randn(B, T, device=device)
        ri = torch.zeros(9, device=device)
        nz = torch.randn(B, T, 9, device=device)
        from ddsp_svc_amd import nsf_source

        def dsp():
            s2 = synth.fast_source(f0, SR, HOP)
            w = synth.combsubsuperfast_synth(f0, s2, ctrls["harmonic_magnitude"], ctrls["harmonic_phase"],
                                             ctrls["noise_magnitude"], ctrls["noise_phase"], gauss, seam.ddsp.window, SR, HOP)
            m = seam.stft.get_mel(w).transpose(1, 2)
            e = nsf_source.sine_source(f0[..., 0], HOP, SR, seam.source.l_linear.weight, seam.source.l_linear.bias, ri, nz)
            return m, e
        for _ in range(3):
            dsp()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(10):
            dsp()
        ev[1].record()
        torch.cuda.synchronize()
    dsp_ms = ev[0].elapsed_time(ev[1]) / 10
    if rank != 0:
        return
    ms = elapsed / steps * 1e3
    emit(({
        "metric": "audio samples/sec, cascade seam (DDSP synth -> log-mel -> denoiser -> NSF source), stand-in networks",
        "value": B * world * T * steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warm,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "main_diff.py seam for B=%d/GPU x %.0f s (F=%d, T=%d): drop-in CombSubSuperFast(win 2048) with a "
                               "stand-in Unit2Control, k_mel, stand-in denoiser, k_sinegen + stand-in generator body; one "
                               "stream" % (B, a.seconds, F, T), "batch_per_gpu": B, "samples_per_utterance": T,
                   "parallelism": "utterance-shard x%d" % world},
        "dsp_kernels_ms": dsp_ms, "dsp_share_of_step": dsp_ms / ms,
        "note": "integration measurement: the stand-in networks are not the reference's (their cost is not the subject); "
                "dsp_kernels_ms is the product's part -- fast_source + stft filter + log-mel + NSF source"}))


def bench_cascade_ref(a, rank, world, device):
    """BASELINE cfg 5 with the REFERENCE's own networks (random weights), when a reference checkout travelled with the run
    (DDSP_REFERENCE_PATH, tools/with_reference.sh): main_diff.py:356-378's loop body for B utterances of 10 s --
        external DDSP model  ddsp/vocoder.py CombSub(256/256/256), the reference's Unit2Control inside     -> patch_reference(): HIP tail
        vocoder.extract      nsf_hifigan/nvSTFT.py STFT.get_mel                                           -> patch_reference_stft(): k_mel
        shallow diffusion    diffusion/vocoder.py Unit2Mel (WaveNet 20 x 384), dpm-solver, k_step 100 / speed-up 10 = 10 steps
        vocoder.infer        nsf_hifigan/models.py Generator (44.1 kHz / hop 512 layout), SourceModuleHnNSF -> patch_reference_source(): k_sinegen
    -- all of it unmodified reference code except the three patched seams.  Reported: the chain per step, and the same chain
    with the seams unpatched (the reference's own DSP under PyTorch-ROCm) on the same GPU."""
    import contextlib
    import importlib
    root = _reference_root()
    if root is None:
        raise SystemExit("--model cascade_ref needs a reference checkout: run under tools/with_reference.sh (DDSP_REFERENCE_PATH)")
    rvoc = _import_reference_vocoder(root)
    from ddsp_svc_amd import vocoder as V, mel as M, nsf_source as NS
    with contextlib.redirect_stdout(sys.stderr):
        dvoc = importlib.import_module("diffusion.vocoder")
        nm = importlib.import_module("nsf_hifigan.models")
        nv = importlib.import_module("nsf_hifigan.nvSTFT")
        from nsf_hifigan.env import AttrDict
    nv.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: M.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax).numpy()
    h = AttrDict({"resblock": "1", "upsample_rates": [8, 8, 2, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4, 4],
                  "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
                  "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "sampling_rate": SR, "num_mels": 128,
                  "n_fft": 2048, "win_size": 2048, "hop_size": HOP, "fmin": 40, "fmax": 16000})
    B = a.batch_per_gpu
    F = int(a.seconds * SR) // HOP + 1
    T = F * HOP
    g = torch.Generator().manual_seed(31 + rank)
    units = torch.randn(B, F, 768, generator=g).to(device)
    vol = (torch.rand(B, F, 1, generator=g) * 0.1).to(device)
    f0 = torch.from_numpy(synthetic_f0(B, F, 1234 + rank)).to(device)

    def build():
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sys.stderr):
            ddsp = rvoc.CombSub(SR, HOP, 256, 256, 256, n_unit=768, n_spk=1).to(device).eval()
            diff = dvoc.Unit2Mel(768, 1, False, 128, 20, 384, 256).to(device).eval()
            gen = nm.Generator(h).to(device).eval()
            gen.remove_weight_norm()
        stft = nv.STFT(SR, 128, 2048, 2048, HOP, 40, 16000)
        return ddsp, diff, gen, stft

    def chain(mods):
        ddsp, diff, gen, stft = mods
        with torch.no_grad():
            wav, _, _ = ddsp(units, f0, vol)                                          # main_diff.py:359
            mel = stft.get_mel(wav, keyshift=a.keyshift).transpose(1, 2)              # :359 (formant_shift_key), diffusion/vocoder.py:147
            out = diff(units, f0, vol, gt_spec=mel, infer=True, infer_speedup=10, method="dpm-solver", k_step=100,
                       use_tqdm=False)                                                # :366-378
            return gen(out.transpose(1, 2), f0[:, :out.size(1), 0])                   # :379, diffusion/vocoder.py:151-153

    def timed(mods, steps, warm):
        for _ in range(warm):
            out = chain(mods)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = chain(mods)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, out
    steps, warm = max(2, a.steps // 25), 1
    # (1) the reference as it is: its own DSP under PyTorch-ROCm
    t_ref, out_ref = timed(build(), steps, warm)
    # (2) the three seams patched
    V.patch_reference()
    M.patch_reference_stft()
    NS.patch_reference_source()
    try:
        mods = build()
        assert type(mods[0]) is V.CombSub
        t_hip, out = timed(mods, steps, warm)
        # the product's share: the DSP pieces alone on resident inputs
        from ddsp_svc_amd import synth
        with torch.no_grad():
            st = synth.phase(f0, SR, HOP)
            ctrls, _ = mods[0].unit2ctrl(units, f0, st.phase_frames, vol)
            u = torch.rand(B, T, device=device)
            ri = torch.zeros(9, device=device)
            nz9 = torch.randn(B, T, 9, device=device)
            src = mods[2].m_source

            def dsp():
                s2 = synth.phase(f0, SR, HOP)
                w = synth.combsub_synth(f0, s2, ctrls["group_delay"], ctrls["harmonic_magnitude"], ctrls["noise_magnitude"], u, SR,
                                        HOP, noise_is_u01=True, want_components=False)[0]
                m = mods[3].get_mel(w, keyshift=a.keyshift)
                e = NS.sine_source(f0[..., 0], HOP, SR, src.l_linear.weight, src.l_linear.bias, ri, nz9)
                return m, e
            dsp_ms = time_alone(dsp, 10)
    finally:
        V.unpatch_reference()
        nv.STFT.get_mel = nv.STFT._reference_get_mel
        del nv.STFT._reference_get_mel
        nm.SourceModuleHnNSF.forward = nm.SourceModuleHnNSF._reference_forward
        del nm.SourceModuleHnNSF._reference_forward
    assert out.shape == out_ref.shape and out.shape[0] == B and torch.isfinite(out).all() and torch.isfinite(out_ref).all()
    if rank != 0:
        return
    emit({
        "metric": "audio samples/sec, main_diff.py end to end with the reference's networks (cfg 5)" +
                  ("" if a.keyshift == 0 else ", formant shift %g semitones" % a.keyshift), "value": B * world * T / t_hip,
        "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": t_hip * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic, random weights",
        "config": {"workload": "main_diff.py:356-379 for B=%d/GPU x %.0f s (F=%d, T=%d): reference CombSub(256/256/256) + Unit2Control, "
                               "STFT.get_mel, Unit2Mel (WaveNet 20x384, dpm-solver 10 steps, k_step 100), NSF-HiFiGAN Generator "
                               "(512 ch, rates 8-8-2-2-2); DSP seams patched to the HIP library" % (B, a.seconds, F, T),
                   "batch_per_gpu": B, "samples_per_utterance": T, "parallelism": "utterance-shard x%d" % world},
        "reference_unpatched_ms_per_step": t_ref * 1e3, "speedup_over_unpatched": t_ref / t_hip,
        "dsp_kernels_ms": dsp_ms, "dsp_share_of_step": dsp_ms / (t_hip * 1e3),
        "note": "integration line: the neural networks are the reference's own PyTorch modules on the same GPU in both rows; "
                "dsp_kernels_ms is the product's part of the patched chain (phase + CombSub tail + log-mel + NSF source)"})


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="combsub", choices=["combsub", "sins", "combsubfast", "combsubsuperfast", "mel", "sinesrc", "rssloss",
                                                         "cascade_seam", "cascade_ref"])
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--keyshift", type=float, default=0.0,
                    help="--model mel / cascade_ref: get_mel's key shift in semitones (the cascade's formant shift, main_diff.py:359)")
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--fir-impl", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed clock ramp-up before the warm-up steps (0 disables)")
    ap.add_argument("--gather", action="store_true", help="also time the optional gather of the waveforms to rank 0")
    ap.add_argument("--only-steps", action="store_true",
                    help="run exactly warmup + steps of the step and nothing else (no clock ramp-up, no separately timed "
                         "kernels, no baselines): what tools/gpu_traffic.sh and the kernel-trace profiles wrap")
    ap.add_argument("--no-module-mode", action="store_true", help="skip the control-mode (i) timing")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the rocprofv3 PMC passes; report the newest committed traffic summary instead")
    ap.add_argument("--no-also", action="store_true", help="skip the Sins (cfg 3) / CombSubSuperFast lines of the N = 1 run")
    ap.add_argument("--no-in-step-trace", action="store_true",
                    help="do not run the kernel-trace pass that measures the dominant kernel inside the step (roofline.frac then uses the stand-alone time)")
    ap.add_argument("--trace-stats-out", default=None,
                    help="write the per-kernel table of the in-step kernel-trace pass there (what profiles/*_kernel_stats.csv holds)")
    ap.add_argument("--cfg4", action="store_true",
                    help="also run BASELINE cfg 4's per-GPU shape (64 utterances per GPU) -- default at --gpus 8")
    ap.add_argument("--cfg4-batch", type=int, default=64, help="utterances per GPU of the cfg-4 line (BASELINE: 64)")
    ap.add_argument("--cfg4-round-steps", type=int, default=10, help="steps per interleaved round of the cfg-4 line")
    ap.add_argument("--cfg4-keep-cache", action="store_true", help="(A/B) do not release the allocator's cached blocks before the cfg-4 line")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg-4 line (part of the default N = 1 and N = 8 runs)")
    ap.add_argument("--no-cfg4-gather", action="store_true", help="cfg-4 line without opening a 1-rank communicator at N = 1")
    ap.add_argument("--no-parity-gate", action="store_true",
                    help="do not hold the timed output against the oracle before printing (A/B runs of broken ablation builds)")
    a = ap.parse_args(argv)
    if a.only_steps:
        a.prewarm_seconds = 0.0

    rank, world, device, comm = setup_ranks(a)
    import torch.distributed as dist

    from ddsp_svc_amd import _ffi, core, synth, sharding

    if a.model in ("sinesrc", "rssloss"):
        (bench_sinesrc if a.model == "sinesrc" else bench_rssloss)(a, rank, world, device)
        return finish_ranks()
    if a.model == "mel":
        bench_mel(a, rank, world, device)
        return finish_ranks()
    if a.model == "cascade_seam":
        bench_cascade_seam(a, rank, world, device)
        return finish_ranks()
    if a.model == "cascade_ref":
        bench_cascade_ref(a, rank, world, device)
        return finish_ranks()

    B = a.batch_per_gpu
    F = int(a.seconds * SR) // HOP + 1              # the reference's frame-count rule (vocoder.py:222)
    T = F * HOP
    n = a.bins
    step, inp = build_step(a.model, B, F, n, device, seed=1234 + rank, fir_impl=a.fir_impl)
    sizes, f0, ctrls, noise, window, win = inp["sizes"], inp["f0"], inp["ctrls"], inp["noise"], inp["window"], inp["win"]

    def fence():
        if world > 1:
            dist.barrier()
        RT.synchronize()

    prewarm(step, a.prewarm_seconds)
    for _ in range(a.warmup):
        out = step()
    fence()
    ev0, ev1 = RT.event(), RT.event()
    t0 = time.perf_counter()
    ev0.record()                                     # the same K steps on the device clock (launch stream)
    for _ in range(a.steps):
        out = step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    events_ms = ev0.elapsed_time(ev1) / a.steps
    if world > 1:
        tt = torch.tensor([elapsed, events_ms], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, events_ms = float(tt[0].item()), float(tt[1].item())
    assert torch.isfinite(out).all()

    gather_ms = None
    if a.gather:                                     # world 1: a 1-rank RCCL communicator, the same call path
        for _ in range(2):
            sharding.gather_utterances(out, B * world, dst=0)
        fence()
        t1 = time.perf_counter()
        full = sharding.gather_utterances(out, B * world, dst=0)
        fence()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del full
    if a.only_steps:
        if rank == 0:
            emit(({"only_steps": True, "model": a.model, "steps": a.steps, "warmup": a.warmup, "n_gpus": world,
                              "ms_per_step": elapsed / a.steps * 1e3, "ms_per_step_events": events_ms}))
        return finish_ranks()

    if a.model in FAST_MODELS:
        report_fast(a, rank, world, B, F, T, sizes, elapsed, gather_ms, f0, ctrls, noise, window, win)
        return finish_ranks()

    # ---- dominant kernel alone: the time-varying FIR (N = 2(n-1) taps), events on the launch stream.  Two forms ship:
    # the FFT-domain block convolution (k_fir_fft, what the step uses for hop 512 / N <= 512) and the direct form on
    # the f32 MFMA pipe (k_fir_mfma, every other shape); both are timed so either roofline can be read. ----
    N = 2 * (n - 1)
    taps = torch.randn(B, F, N, device=device) / N ** 0.5
    y = torch.empty(B, T, device=device)
    L = _ffi.lib()
    st_ptr = torch.cuda.current_stream().cuda_stream

    def time_fir(impl, reps=20):
        def once():
            return L.ddsp_hip_fft_convolve(noise.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None,
                                           B, F, HOP, N, impl, st_ptr)
        if once() != 0:
            return None
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        # events around a loop of back-to-back launches: the average includes the gap between dependent launches (the
        # write-back of the previous output, ~10 us), so it reads ~10 % above the kernel-trace duration of the same
        # kernel (profiles/*_kernel_stats.csv) -- the conservative side for the roofline fraction
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    used_impl = a.fir_impl if a.fir_impl else (5 if N <= 512 else (4 if N <= 1022 else 3))      # what launch_fir's AUTO picks at hop 512
    fir_ms = time_fir(used_impl)
    mfma_ms = time_fir(3) if used_impl != 3 else fir_ms
    fir_flops = 4.0 * N * B * T                      # direct form: 2N multiply-adds per output sample
    fir_launches = 3 if a.model == "combsub" else 2
    fir_bytes = (8.0 + 4.0 * N / HOP) * B * T        # input + output + one tap row per frame
    kname = {4: "k_fir_fft", 5: "k_fir_blk6"}.get(used_impl, "k_fir_mfma")      # (the product library ships one generation per kernel)
    # BASELINE cfg 4's per-GPU shape: every rank takes part (collectives inside)
    # (every rank takes part; at N = 1 it is part of the default command, so that the driver's run records the per-GPU shape
    # every multi-GPU point is made of)
    want_cfg4 = a.model == "combsub" and not a.no_cfg4 and (world == 8 or a.cfg4 or (world == 1 and (B, F, n) == (32, 862, 256)
                                                                                   and not a.no_also))
    cfg4 = cfg4_line(a, rank, world, device, F, n, comm) if want_cfg4 else None
    # parity gate: rank 0's timed output against the oracle BEFORE any number is printed
    parity, gate_fail = None, None
    if rank == 0 and not a.no_parity_gate:
        try:
            parity = parity_gate(a.model, f0, ctrls, noise, out)
        except SystemExit as e:
            gate_fail = str(e)
    if world > 1:                                    # the verdict reaches every rank: nobody is left waiting in a collective
        flag = torch.tensor([1.0 if gate_fail else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) != 0.0:
            finish_ranks()
            raise SystemExit(gate_fail or "bench.py: rank 0's parity gate failed")
    elif gate_fail:
        raise SystemExit(gate_fail)
    # arithmetic the FFT forms actually execute (5 N log2 N per complex transform + spectral products): per frame pair
    # three 2048-point transforms (k_fir_fft) or per hop-block pair four 1024-point ones (k_fir_blk)
    if used_impl == 5:
        fft_flops = B * ((F + 1) // 2) * (4 * 5.0 * 1024 * 10 + 1024 * 3 * 14.0)
    else:
        fft_flops = B * ((F + 2) // 2) * (3 * 5.0 * 2048 * 11 + 2048 * 2 * 10.0)

    if rank == 0:
        total = B * world * T * a.steps
        value = total / elapsed
        ms = elapsed / a.steps * 1e3
        sigma_c = sum(sizes)
        headline_shape = (B, F, n) == (32, 862, 256)
        want_live = world == 1 and headline_shape and not a.no_live_traffic
        live = live_traffic(a.model, ("--fir-impl", str(a.fir_impl)) if a.fir_impl else ()) if want_live else None
        traffic, tr, traffic_source = traffic_of(kname, a.model, live) if headline_shape else (None, None, None)
        alg_bytes = (8.0 + 4.0 * (sigma_c + 1) / HOP) * B * T          # out + noise + controls + f0
        # the dominant kernel INSIDE the step at the clocks' steady state (a kernel-trace pass of this run): what `roofline.frac` is
        # made of.  SURVEY 8-d bytes of ONE filter: input + output + its n control bins per frame (the tap rows are an intermediate,
        # not algorithmic bytes); a launch of the fused layout carries one or two filters, so per launch: x filters / launches.
        want_trace = world == 1 and not a.no_live_traffic and not a.no_in_step_trace
        instep = in_step_kernel_times(a.model, ("--fir-impl", str(a.fir_impl)) if a.fir_impl else (), stats_out=a.trace_stats_out) \
            if want_trace else None
        ks = next((v for k_, v in (instep or {}).items() if k_.startswith(kname.rstrip("<"))), None)
        filt_bytes = (8.0 + 4.0 * n / HOP) * B * T
        if ks:
            per_launch = fir_launches / ks["launches_per_step"]          # filters per launch (1.5 in the fused CombSub step)
            k_us, k_src = ks["avg_us"], "in-step: rocprofv3 --kernel-trace pass of this run, the last %d of %d launches" % (ks["launches_averaged"], ks["launches_traced"])
        else:
            per_launch, k_us, k_src = 1.0, fir_ms * 1e3, "alone (no in-step trace in this run): HIP events around 20 back-to-back launches"
        k_bytes, k_flops = filt_bytes * per_launch, fft_flops * per_launch
        res = {
            "metric": "audio samples/sec, CombSub 44.1kHz 256-harm hop512" if a.model == "combsub"
                      else "audio samples/sec, Sins 44.1kHz 256-harm hop512",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "prewarm_s": a.prewarm_seconds,
            "ms_per_step": ms, "ms_per_step_events": events_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "config": {"workload": "%s B=%d/GPU x %.0f s utterances (F=%d, T=%d), n_mag %d/%d/%d, sr 44100, hop 512, "
                                   "DSP path (HOT-1 + HOT-2) from resident f0 / raw controls ~N(0,1) / uniform noise, "
                                   "signal only" % (a.model, B, a.seconds, F, T, n, n, n),
                       "batch_per_gpu": B, "frames": F, "samples_per_utterance": T, "parallelism": "utterance-shard x%d" % world},
            "roofline": {"kernel": kname.rstrip("<"), "bound": "valu",
                         "achieved": k_bytes / (k_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": k_bytes / (k_us * 1e-6) / 1e9 / 8000.0,
                         "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": k_bytes, "algorithmic_bytes_per_filter": filt_bytes,
                         "avg_us": k_us, "avg_source": k_src, "filters_per_step": fir_launches,
                         "launches_per_step": ks["launches_per_step"] if ks else fir_launches, "filters_per_launch": per_launch,
                         "valu_frac": k_flops / (k_us * 1e-6) / 1e12 / 157.3, "valu_peak_TFLOPs": 157.3,
                         "frac_alone": filt_bytes / (fir_ms * 1e-3) / 1e9 / 8000.0, "avg_ms_alone": fir_ms, "avg_ms": fir_ms,
                         "step_hbm_frac": alg_bytes / (ms * 1e-3) / 1e9 / 8000.0,
                         "step_traffic_ratio": (tr["hbm_bytes"] / alg_bytes) if tr else None,
                         # the bytes the step MOVES (PMC, intermediates included) over its time, against 8 TB/s: how close the step
                         # as a whole sits to the memory roof (its kernels are issue-bound one by one, the step is not far from both)
                         "step_traffic_hbm_frac": (tr["hbm_bytes"] / (ms * 1e-3) / 1e9 / 8000.0) if tr else None,
                         "in_step_kernels": instep,
                         "note": "frac = SURVEY 8-d algorithmic HBM bytes of what one launch of the dominant kernel processes / its "
                                 "average duration INSIDE the step (steady state) / 8 TB/s.  `bound` names the roof that binds the kernel: "
                                 "the vector ALU's issue rate (valu_frac = executed FFT flops / time / 157.3 TFLOP/s; butterflies are "
                                 "add / sub / mul, no FMA: at most half of that peak is attainable), not HBM -- a fused DSP path is at "
                                 "0.1 - 0.3 of the HBM roof by construction (SURVEY 8-d).  frac_alone: one filter timed alone (HIP events, "
                                 "20 back-to-back launches).  step_hbm_frac / step_traffic_ratio: the WHOLE step's algorithmic bytes over "
                                 "its wall time, and its PMC bytes over its algorithmic bytes -- the path's HBM numbers; frac is one "
                                 "kernel's."},
            "roofline_compute": (
                {"kernel": kname.rstrip("<"), "bound": "valu", "unit": "TFLOP/s", "peak": 157.3,
                 "achieved": fft_flops / (fir_ms * 1e-3) / 1e12, "frac": fft_flops / (fir_ms * 1e-3) / 1e12 / 157.3,
                 "executed_flops_per_launch": fft_flops,
                 "direct_form_equivalent_TFLOPs": fir_flops / (fir_ms * 1e-3) / 1e12,
                 "note": "vector f32 peak counts packed FMA; FFT butterflies are add/sub/mul (no FMA) so the attainable "
                         "issue rate is at most half of it"} if used_impl in (4, 5) else
                {"kernel": "k_fir_mfma", "bound": "mfma", "unit": "TFLOP/s", "peak": 157.3,
                 "achieved": fir_flops / (fir_ms * 1e-3) / 1e12, "frac": fir_flops / (fir_ms * 1e-3) / 1e12 / 157.3,
                 "algorithmic_flops_per_launch": fir_flops}),
            "roofline_mfma_direct_form": {"kernel": "k_fir_mfma", "bound": "mfma", "unit": "TFLOP/s", "peak": 157.3,
                                          "achieved": fir_flops / (mfma_ms * 1e-3) / 1e12 if mfma_ms else None,
                                          "frac": fir_flops / (mfma_ms * 1e-3) / 1e12 / 157.3 if mfma_ms else None,
                                          "avg_ms": mfma_ms, "algorithmic_flops_per_launch": fir_flops,
                                          "note": "the direct form on v_mfma_f32_16x16x4_f32, used for shapes outside the "
                                                  "FFT form; ~2.05 GHz under load, 1.29x issued/algorithmic MFMAs"},
            "roofline_step_hbm": {"bound": "hbm", "algorithmic_bytes_per_step": alg_bytes,
                                  "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": alg_bytes / (ms * 1e-3) / 1e9 / 8000.0},
        }
        res.update(comm)
        res["parity_vs_oracle"] = parity
        res["roofline_step_traffic"] = None if tr is None else {
            "pmc_bytes_per_step": tr["hbm_bytes"], "algorithmic_bytes_per_step": alg_bytes,
            "ratio": tr["hbm_bytes"] / alg_bytes, "launches_per_step": tr.get("launches_per_step"), "source": tr["source"],
            "traffic_source": traffic_source}
        if cfg4 is not None:
            res["cfg4"] = cfg4
        if world == 1 and a.model == "combsub" and not a.no_also:
            res["also"] = also_lines(a, device, B, F, n, want_live)
        if gather_ms is not None:
            res["gather_ms"] = gather_ms
            res["gather_bytes_per_rank"] = 4.0 * B * T
        if world == 1 and not a.no_module_mode:
            res["value_module_mode"] = module_mode(a.model, B, F, n, device, max(5, a.steps // 5), 3)
            # DSP-only step with the uniform noise drawn inside the noise filter (opt-in; algorithmic bytes 10.01 / sample)
            fn = synth.combsub_synth if a.model == "combsub" else synth.sins_synth

            def step_rng():
                st = synth.phase(f0, SR, HOP)
                return fn(f0, st, ctrls[0], ctrls[1], ctrls[2], None, SR, HOP, want_components=False, noise_seed=7)[0]
            prewarm(step_rng, min(a.prewarm_seconds, 0.3))         # (the module-mode steps before this were another workload: the clocks settle again)
            torch.cuda.synchronize()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record()
            for _ in range(a.steps):
                o2 = step_rng()
            r1.record()
            torch.cuda.synchronize()
            assert torch.isfinite(o2).all()
            res["in_kernel_noise"] = {"ms_per_step": r0.elapsed_time(r1) / a.steps,
                                      "algorithmic_bytes_per_step": (4.0 + 4.0 * (sigma_c + 1) / HOP) * B * T,
                                      "note": "opt-in: Philox4x32-10 draw inside the noise filter instead of a resident "
                                              "noise tensor (a stream of its own, not torch.rand's)"}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.model, F, sizes)
            res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
            if _reference_root():                        # a reference checkout travelled with this run: it IS the baseline
                res["cpu_baseline_port"] = res["cpu_baseline"]
                res["cpu_baseline"] = cpu_baseline_reference(a.model, F, n, device)
                res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
            res["cpu_baseline_aten_chain"] = cpu_baseline_aten_chain(a.model, F, sizes)
            res["cpu_baseline_aten_chain"]["gpu_over_cpu"] = value / res["cpu_baseline_aten_chain"]["value"]
            if a.model in ("combsub", "sins"):
                try:
                    g_ = aten_chain_on_gpu(a.model, f0, ctrls, noise, out, device)
                    g_["hip_over_eager"] = value / g_["value"]
                except Exception as e:                                  # a baseline that cannot RUN must not take the line down
                    g_ = {"error": "%s: %s" % (type(e).__name__, e)}
                res["cpu_baseline_aten_chain"]["same_chain_on_gpu"] = g_
                # ... but one that ran is a second, full-batch parity gate: the reference's op chain on the same GPU
                if not a.no_parity_gate and g_.get("rel_rms_vs_hip_path") is not None and not g_["rel_rms_vs_hip_path"] <= 1e-5:
                    raise SystemExit("bench.py: the timed output differs from the reference's op chain on the same inputs: "
                                     "rel RMS %.3e > 1e-5" % g_["rel_rms_vs_hip_path"])
                g_["gate"] = "rel_rms_vs_hip_path <= 1e-5 or no line"
        emit((res))
    finish_ranks()


if __name__ == "__main__":
    try:
        main()
    finally:
        flush_emit()
