"""The C ABI promises re-entrancy across streams and host threads (include/zedo_hip.h, SURVEY.md section 8b: "re-entrant across
distinct streams"); the reference's own callers are single-threaded on the default stream (run/opt_main.py:202-220 is the
caller this generalises).  Here two host threads, each on its own non-default stream, push two DIFFERENT problems through
zedo_ipo_fit -> zedo_rotate_init -> zedo_oil_run -> zedo_min_mpjpe at the same time, with the sampling profiler switched on -
once through the Python binding (per-call workspaces from torch's stream-aware allocator) and once through raw ctypes calls with
caller-owned workspaces - and every repetition must reproduce the serial default-stream results bit for bit."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REPS = 20


@pytest.fixture(scope="module")
def zh():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import zedo_hip
    return zedo_hip


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


class Problem:
    """One small ZeDO problem resident on the device (inputs only)."""

    def __init__(self, zh, W, N, H, keylist, ipo_T, min_s, seed, S=40, ipo_it=60):
        import zedo_oracle as O
        from lib.dataset import synthetic as syn
        d = syn.make_poses(N, seed=seed, conf_mode="uniform")
        cl = syn.make_clusters(H, seed=seed)
        self.N, self.H, self.S, self.B = N, H, S, N * H
        self.keylist, self.ipo_T, self.min_s, self.ipo_it = list(keylist), ipo_T, min_s, ipo_it
        self.x0 = dev(cl - cl[:, 0:1])
        self.uv = dev(d["db_2d"][:, :, :2])
        self.K = dev(d["camera_param"])
        self.geom = zh.reproj_prepare(self.uv, self.K, dev(d["db_2d"][:, :, 2]))
        gt = d["db_3d"].astype(np.float64)
        self.gt = dev(gt - gt[:, 0:1], torch.float64)
        self.sched = zh.Schedule(W, O.oil_timestamps(S))
        self.W = W

    def run_binding(self, zh):
        R, T = zh.ipo_fit(self.x0, self.uv, self.K, self.keylist, "z", self.ipo_T, self.min_s, 2.0, self.ipo_it,
                          self.N * len(self.keylist) * 2, self.B)
        x = zh.rotate_init(self.x0, R, self.N)
        zh.oil_run(self.W, self.sched, x, self.geom, T, 0, self.S, self.S // 5)
        e1, b1, i1 = zh.min_mpjpe(x, self.gt, self.N, False)
        e2, b2, i2 = zh.min_mpjpe(x, self.gt, self.N, True)
        return [R, T, x, e1, b1, i1, e2, b2, i2]

    def raw_buffers(self, zh):
        B, N = self.B, self.N
        f32, f64, i32 = torch.float32, torch.float64, torch.int32
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device="cuda")
        return dict(R=e((B, 3, 3), f32), T=e((B, 3), f32), x=e((B, 17, 3), f32), e1=e((B,), f64), b1=e((N,), f64), i1=e((N,), i32),
                    e2=e((B,), f64), b2=e((N,), f64), i2=e((N,), i32), ws=e((zh.workspace_bytes(B),), torch.uint8))

    def run_raw(self, zh, buf, stream):
        """The same chain through the C ABI itself: caller-owned outputs and workspace, an explicit hipStream_t."""
        lib, P = zh._lib, lambda t: ctypes.c_void_p(t.data_ptr())
        st = ctypes.c_void_p(stream.cuda_stream)
        kl = (ctypes.c_int * len(self.keylist))(*self.keylist)
        B, N, H = self.B, self.N, self.H
        rc = lib.zedo_ipo_fit(P(self.x0), P(self.uv), P(self.K), ctypes.cast(kl, ctypes.c_void_p), len(self.keylist), 4, self.ipo_T,
                              self.min_s, 2.0, self.ipo_it, float(N * len(self.keylist) * 2), P(buf["R"]), P(buf["T"]), None, None,
                              B, H, N, 17, 0, st)
        assert rc == 0
        assert lib.zedo_rotate_init(P(self.x0), P(buf["R"]), P(buf["x"]), B, H, N, 17, 0, st) == 0
        assert lib.zedo_oil_run(self.W._h, self.sched._h, P(buf["x"]), P(self.geom), P(buf["T"]), 0, self.S, self.S // 5, B, N, 0,
                                P(buf["ws"]), buf["ws"].numel(), st) == 0
        for pr, (e, b, i) in ((0, ("e1", "b1", "i1")), (1, ("e2", "b2", "i2"))):
            assert lib.zedo_min_mpjpe(P(buf["x"]), P(self.gt), B, N, 17, 0, pr, P(buf[e]), P(buf[b]), P(buf[i]), st) == 0
        return [buf[k] for k in ("R", "T", "x", "e1", "b1", "i1", "e2", "b2", "i2")]


@pytest.fixture(scope="module")
def problems(zh, weights0, math_mode):
    W = zh.Weights(weights0)
    probs = [Problem(zh, W, 48, 3, range(17), 8.0, 0.2, seed=21),          # 3DPW settings (configs/optim/*_pw3d.py)
             Problem(zh, W, 80, 2, [0, 1, 4], 3.0, 0.5, seed=22)]          # H36M settings
    torch.cuda.synchronize()
    serial = [[t.clone() for t in p.run_binding(zh)] for p in probs]       # default stream, one after the other
    torch.cuda.synchronize()
    return probs, serial


def _run_threads(workers):
    errs = []

    def guard(fn):
        def w():
            try:
                fn()
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
        return w
    ts = [threading.Thread(target=guard(fn)) for fn in workers]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not any(t.is_alive() for t in ts), "a worker thread hung"
    if errs:
        raise errs[0]


@pytest.mark.parametrize("route", ["binding", "raw"])
def test_two_threads_two_streams_equal_the_serial_runs(zh, problems, route):
    probs, serial = problems
    gate = threading.Barrier(2, timeout=300)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [p.raw_buffers(zh) for p in probs] if route == "raw" else [None, None]
    torch.cuda.synchronize()
    mismatches = []
    zh.profile_start(sample_every=3, max_samples=4096)

    def worker(k):
        def body():
            p, s = probs[k], streams[k]
            for rep in range(REPS):
                gate.wait()                                  # both threads enter every repetition together
                if route == "binding":
                    with torch.cuda.stream(s):               # thread-local: the other thread keeps its own stream
                        out = p.run_binding(zh)
                else:
                    out = p.run_raw(zh, bufs[k], s)
                s.synchronize()
                for name, got, ref in zip("R T x e1 b1 i1 e2 b2 i2".split(), out, serial[k]):
                    if not torch.equal(got, ref):
                        mismatches.append((k, rep, name))
        return body

    _run_threads([worker(0), worker(1)])
    prof = zh.profile_stop()
    assert not mismatches, f"concurrent runs differ from the serial ones: {mismatches[:8]}"
    # the profiler saw the launches of BOTH streams: 4 hidden + 1 pre + 1 post launch per OIL iteration and problem
    steps = REPS * sum(p.S for p in probs)
    assert prof["hidden_dense"]["launches"] == 4 * steps and prof["pre_dense"]["launches"] == steps
    assert prof["hidden_dense"]["samples"] > 0 and prof["hidden_dense"]["avg_ms_raw"] > 0


def test_a_side_stream_sees_no_default_stream_state(zh, problems):
    """The binding takes the stream of the TENSORS' device at call time (it used to take the current device's) and a
    workspace per call: the same call on a side stream, on the default stream and interleaved gives the same bits."""
    probs, serial = problems
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        a = probs[0].run_binding(zh)
    b = probs[1].run_binding(zh)                               # default stream, overlapping the side stream's work
    with torch.cuda.stream(s):
        c = probs[1].run_binding(zh)
    torch.cuda.synchronize()
    for got, ref in ((a, serial[0]), (b, serial[1]), (c, serial[1])):
        assert all(torch.equal(g, r) for g, r in zip(got, ref))
    with pytest.raises(zh.ZedoError):                          # host tensors (another "device") are refused, not copied
        zh.min_mpjpe(a[2].cpu(), probs[0].gt, probs[0].N)


def test_the_whole_chain_is_capturable_into_a_hip_graph(zh, problems):
    """No entry point of the chain synchronises or copies: zedo_ipo_fit included (Adam's bias-correction table is part of the
    code object since round 5; rounds 1-4 uploaded it with a blocking copy inside the first fit of a device)."""
    probs, serial = problems
    p = probs[0]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        p.run_binding(zh)                                      # warm the allocator on the capture stream
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = p.run_binding(zh)
    for _ in range(3):
        for t in out:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(got, ref) for got, ref in zip(out, serial[0]))
