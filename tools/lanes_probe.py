"""Where a split call (csrc/api.hip, lanes) differs from the unsplit one at full size: per output, per utterance, with two lanes
and with one (a difference that only two lanes show is a race between the lanes; one that both show is the kernels' own
dependence on the launch geometry)."""
import sys

import torch

sys.path.insert(0, ".")
from ddsp_svc_amd import _ffi, synth  # noqa: E402
from tests.test_lanes import _combsub_inputs, _combsub  # noqa: E402

dev = torch.device("cuda:0")
for B in (32, 64):
    _, tensors = _combsub_inputs(B, 862, dev, seed=B)
    _ffi.set_tuning("LANE_ROWS", 1)
    whole = _combsub(tensors)
    whole2 = _combsub(tensors)
    print("B", B, "unsplit repeat equal:", [bool(torch.equal(a, b)) for a, b in zip(whole, whole2)])
    for lanes in (2, 1):
        _ffi.set_tuning("LANE_ROWS", 0 if len(sys.argv) < 2 else int(sys.argv[1]))
        _ffi.set_tuning("LANES", lanes)
        for rep in range(2):
            split = _combsub(tensors)
            torch.cuda.synchronize()
            for name, a, b in zip(("signal", "harmonic", "noise"), whole, split):
                d = (a - b).abs()
                bad = (d.amax(dim=1) > 0).nonzero().flatten().tolist()
                first = [(int(u), int((d[u] > 0).nonzero()[0]), int((d[u] > 0).sum())) for u in bad[:4]]
                print("  lanes", lanes, "rep", rep, name, "max", float(d.max()), "utterances", bad[:12], "(utt, first sample, count)", first)
    _ffi.set_tuning("LANES", 0)
    _ffi.set_tuning("LANE_ROWS", 0)
