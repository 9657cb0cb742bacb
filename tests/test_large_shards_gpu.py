"""Per-GPU shards of the 8-GPU BASELINE configurations on ONE MI355X (-m gpu): configs[4] (100 000 poses x H = 50
over 8 GPUs = 625 000 rows per GPU), a 1 200 000-row shard that crosses the default 2^20-row chunk boundary of
zedo_oil_run / zedo_sde_step, and configs[3]'s own shard: 567 040 poses x H = 50 over 8 GPUs = 3 544 000 rows per GPU,
walked in four chunks (three seams).  Size-independent
properties: every row is independent of the batch it travels in, so any slice recomputed on its own (with its
row_offset) must reproduce the full run BIT FOR BIT - at the first rows, across the chunk seam, and in the last
chunk - through IPO, the fixed-T steps, the switch to the least-squares T, and the selection."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CHUNK = 1 << 20


@pytest.fixture(scope="module")
def zh():
    import zedo_hip
    return zedo_hip


@pytest.mark.parametrize("N,H", [(12500, 50), (24000, 50), (70880, 50)],
                         ids=["625000_rows_config4_shard", "1200000_rows_two_chunks", "3544000_rows_config3_shard_four_chunks"])
def test_large_shard_slices_are_bitwise_the_full_run(zh, weights0, N, H):
    import zedo_oracle as O
    from lib.dataset import synthetic as syn
    B = N * H
    d = syn.make_poses(N, seed=404, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=404)
    dev = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    W = zh.Weights(weights0)
    S = 1000
    sched = zh.Schedule(W, O.oil_timestamps(S))
    uv, K, conf = dev(d["db_2d"][:, :, :2]), dev(d["camera_param"]), dev(d["db_2d"][:, :, 2])
    geom = zh.reproj_prepare(uv, K, conf, conf)
    x0 = dev(cl - cl[:, 0:1])
    kl, norm = list(range(17)), N * 17 * 2
    assert zh.workspace_bytes(B) == min(B + (-B) % 64, CHUNK) * (64 + 2048) * 4          # bounded by one chunk
    R, T0 = zh.ipo_fit(x0, uv, K, kl, "z", 8.0, 0.2, 2.0, 500, norm, B)
    xi = zh.rotate_init(x0, R, N)
    x, T = xi.clone(), T0.clone()
    lo_step, hi_step, switch = 198, 201, 200            # two steps with the IPO translation, one with the solved one
    zh.oil_run(W, sched, x, geom, T, lo_step, hi_step, switch)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(T).all())
    assert not torch.equal(T, T0)                        # the least-squares T replaced the IPO one
    slices = [(0, 300), (B - 333, B), (B // 2 - 77, B // 2 + 1000)]
    for seam in range(CHUNK, B, CHUNK):                  # every chunk seam of the walk
        slices += [(seam - 300, seam + 300), (seam, seam + 256), (seam - 1, seam + 1)]
    assert len(slices) == 3 + 3 * ((B - 1) // CHUNK)
    for a, b in slices:
        Rs, Ts = zh.ipo_fit(x0, uv, K, kl, "z", 8.0, 0.2, 2.0, 500, norm, b - a, row_offset=a)
        assert torch.equal(Rs, R[a:b]) and torch.equal(Ts, T0[a:b]), ("ipo", a, b)
        xs = zh.rotate_init(x0, Rs, N, row_offset=a)
        assert torch.equal(xs, xi[a:b]), ("rotate", a, b)
        zh.oil_run(W, sched, xs, geom, Ts, lo_step, hi_step, switch, row_offset=a)
        assert torch.equal(xs, x[a:b]) and torch.equal(Ts, T[a:b]), ("oil", a, b)
    # per-step surface over the same rows (zedo_sde_step walks the chunks too)
    y = x.clone()
    zh.sde_step(W, sched, 500, y)
    a, b = (CHUNK - 128, CHUNK + 128) if B > CHUNK else (B - 256, B)
    ys = x[a:b].clone()
    zh.sde_step(W, sched, 500, ys)
    assert torch.equal(ys, y[a:b])
    # selection over the whole shard equals the selection over its two halves combined (first minimum wins)
    gt = dev((d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64), torch.float64)
    for p2 in (False, True):
        err, best, idx = zh.min_mpjpe(x, gt, N, procrustes=p2)
        cut = (H // 2) * N + 4321
        _, b1, i1 = zh.min_mpjpe(x[:cut].contiguous(), gt, N, procrustes=p2)
        _, b2, i2 = zh.min_mpjpe(x[cut:].contiguous(), gt, N, procrustes=p2, row_offset=cut)
        gb = torch.minimum(b1, b2)
        gi = torch.where(b1 <= b2, i1, i2)
        assert torch.equal(gb, best) and torch.equal(gi, idx)
        assert torch.equal(best, err.reshape(H, N).min(0).values)


def test_eight_virtual_ranks_reproduce_the_unsharded_run(zh, weights0):
    """BASELINE configs[3] / [4] shard the hypothesis-major rows over 8 GPUs (shard_rows, the rule of
    lib/dataset/EvaSampler.py:78-107) and combine the per-pose minima with MIN all-reduces.  One GPU plays the 8
    ranks in turn: every rank's Pipeline.run(row_offset, rows) must give exactly its rows of the unsharded run,
    and the per-rank selections combined the way reduce_min_over_ranks combines them (minimum, then lowest
    hypothesis index among the holders) must equal the unsharded selection - IPO normaliser, row_offset
    arithmetic, uneven last shard and shard boundaries inside a hypothesis included (H*N = 50*2003 rows)."""
    from lib.dataset import synthetic as syn
    from zedo_hip.pipeline import Pipeline, ZeDOConfig, shard_rows
    N, H, S, world = 2003, 50, 12, 8
    d = syn.make_poses(N, seed=808, conf_mode="uniform")
    cl = syn.make_clusters(H, seed=808)
    pipe = Pipeline(weights0, ZeDOConfig.pw3d(OIL_iterations=S), "cuda").load(cl, d["db_2d"], d["camera_param"])
    x_full, T_full = pipe.run()
    gt = (d["db_3d"] - d["db_3d"][:, 0:1]).astype(np.float64)
    sel_full = pipe.select(x_full, gt)
    big = 2 ** 31 - 1
    acc = {k: (torch.full((N,), float("inf"), dtype=torch.float64, device="cuda"),
               torch.full((N,), big, dtype=torch.int64, device="cuda")) for k in ("p1", "p2")}
    parts = []
    covered = 0
    for rank in range(world):
        lo, rows = shard_rows(H * N, rank, world)
        assert lo == covered
        covered += rows
        x, T = pipe.run(row_offset=lo, rows=rows)
        assert torch.equal(x, x_full[lo:lo + rows]) and torch.equal(T, T_full[lo:lo + rows]), rank
        parts.append((lo, rows))
        sel = pipe.select(x, gt, row_offset=lo)
        for k in ("p1", "p2"):
            b, i = sel[k]
            i = torch.where(i < 0, torch.full_like(i, big), i).long()
            gb, gi = acc[k]
            nb = torch.minimum(gb, b)
            ni = torch.minimum(torch.where(gb == nb, gi, torch.full_like(gi, big)),
                               torch.where(b == nb, i, torch.full_like(i, big)))
            acc[k] = (nb, ni)
    assert covered == H * N and len({r for _, r in parts}) <= 2          # contiguous, unpadded, at most one short shard
    for k in ("p1", "p2"):
        assert torch.equal(acc[k][0], sel_full[k][0]) and torch.equal(acc[k][1], sel_full[k][1].long()), k
