"""The 8-GPU prediction of profiles/r04/scale_model.md is reproducible from the committed traces (CPU only): scripts/scale_model.py
replays rank_factor()'s three-queue schedule with the per-phase durations measured on the real kernels and a link model."""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
TRACES = os.path.join(ROOT, "profiles", "r04", "scale_traces_N65536.json.gz")


def test_replay_of_the_committed_traces():
    import scale_model as sm
    res = json.load(gzip.open(TRACES, "rt"))
    n, t1 = res["n"], res["single_gpu_s"]
    cfg = {(c["Pr"], c["Pc"], c["nb"]): c for c in res["configs"] if "trace" in c}
    # the replay itself: one device, no transfers -- must reproduce the measured wall clock
    one = cfg[(1, 1, 1024)]
    assert abs(sm.replay(one, n, 60.0, 25.0) / one["wall_full_s"] - 1.0) < 0.05
    # whole tile rows per rank (8 x 1, snake): inside the 6x budget at every link rate assumed, and bound by the updates, not the chain
    best = cfg[(8, 1, 1024)]
    for bw in (45.0, 60.0, 75.0):
        assert sm.replay(best, n, bw, 25.0) < t1 / 6.0
    assert sm.replay(best, n, 60.0, 25.0, chain_only=True) < 0.5 * sm.replay(best, n, 60.0, 25.0)
    # round 3's 2 x 4 grid: the row panel on one link inside the chain -- outside the budget, and sensitive to the link rate
    old = cfg[(2, 4, 1024)]
    assert sm.replay(old, n, 60.0, 25.0) > t1 / 6.0
    assert sm.replay(old, n, 45.0, 25.0) > 1.1 * sm.replay(old, n, 75.0, 25.0)
    # every traced run computed the right factor
    assert all(c["traced_logdet_rel"] < 1e-12 for c in cfg.values())


def test_snake_order_balances_the_lower_triangle():
    import scale_model as sm
    for W, nt in ((8, 64), (8, 128), (4, 48), (3, 30)):
        share = [0] * W
        for i in range(nt):
            share[sm.prow(i, W, True)] += i + 1
        assert max(share) / (sum(share) / W) < 1.04
        plain = [0] * W
        for i in range(nt):
            plain[i % W] += i + 1
        assert max(plain) / (sum(plain) / W) > max(share) / (sum(share) / W)
