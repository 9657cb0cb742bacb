#!/usr/bin/env python3
"""Package power and clocks of the MI355X under the loads of this path: is the 1.5-1.6 GHz the split-fp16
hidden layers run at a POWER cap?  A sampler thread polls `rocm-smi --showpower --showclocks --showmaxpower --json` (every ~0.25 s)
while the main thread runs, one after the other: idle, the bare fp32 MFMA probe, the bare fp16 MFMA probe (zedo_probe_mfma_peak[_f16]),
the OIL loop at BASELINE configs[2]'s 50 750 rows in the exact-fp32 mode and in the split-fp16 mode.  Output: one JSON document
(phases with their samples, per-phase mean / max power, the in-kernel clock of each probe).

    python tools/power_probe.py gpurun_out/power_r06.json
"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zedo-release_amd"))


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10)
        return json.loads(r.stdout) if r.stdout.strip().startswith("{") else {"raw": r.stdout[-400:], "err": r.stderr[-400:]}
    except Exception as e:       # noqa: BLE001 - the tool reports whatever it could not read
        return {"error": repr(e)}


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.phase, self.stop = [], "start", False

    def run(self):
        while not self.stop:
            t = time.time()
            self.samples.append(dict(t=t, phase=self.phase, smi=smi()))
            time.sleep(max(0.0, 0.25 - (time.time() - t)))


def power_of(sample):
    for card in sample.get("smi", {}).values():
        if isinstance(card, dict):
            for k, v in card.items():
                if "Power (W)" in k and "Max" not in k:
                    try:
                        return float(v)
                    except (TypeError, ValueError):
                        pass
    return None


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/power_probe.json"
    import numpy as np
    import torch
    import zedo_hip as zh
    from lib.dataset import synthetic as syn
    from zedo_hip.pipeline import Pipeline, ZeDOConfig
    torch.cuda.set_device(0)
    sm = Sampler()
    sm.start()
    phases = {}

    def phase(name, fn, seconds):
        sm.phase = name
        t0, info = time.time(), []
        while time.time() - t0 < seconds:
            info.append(fn())
        torch.cuda.synchronize()
        phases[name] = info

    phase("idle", lambda: time.sleep(0.5), 2.0)
    phase("probe_fp32_mfma", lambda: zh.probe_mfma_peak(400000), 4.0)
    phase("probe_fp16_mfma", lambda: zh.probe_mfma_peak_f16(2000000), 4.0)
    w = syn.make_weights(seed=0)
    d = syn.make_poses(1015, seed=2024)
    pipe = Pipeline(w, ZeDOConfig.pw3d(OIL_iterations=1000), "cuda:0").load(syn.make_clusters(50, seed=2024), d["db_2d"], d["camera_param"])
    for math in ("f32", "f16x3"):
        pipe.weights.set_math(math)
        pipe.run(oil_steps=20)

        def one(m=math):
            zh.profile_start(sample_every=37, max_samples=512)
            t0 = time.time()
            pipe.run()
            torch.cuda.synchronize()
            dt = time.time() - t0
            pr = zh.profile_stop()["hidden_dense"]
            return dict(pass_s=round(dt, 4), hidden_ms=round(pr["avg_ms"], 4), kernel_ghz=round(pr.get("shader_clock_ghz") or 0.0, 3))
        phase("oil_" + math, one, 8.0)
    phase("idle_after", lambda: time.sleep(0.5), 2.0)
    sm.stop = True
    sm.join(timeout=5)
    summary = {}
    for name in phases:
        pw = [p for p in (power_of(s) for s in sm.samples if s["phase"] == name) if p is not None]
        summary[name] = dict(samples=len(pw), power_w_mean=round(float(np.mean(pw)), 1) if pw else None, power_w_max=max(pw) if pw else None,
                             results=phases[name][-3:] if name.startswith(("probe", "oil")) else None)
    first = next((s["smi"] for s in sm.samples if isinstance(s.get("smi"), dict)), {})
    doc = dict(tool="tools/power_probe.py", summary=summary, first_smi_sample=first, samples=[dict(t=round(s["t"] - sm.samples[0]["t"], 2), phase=s["phase"],
                                                                                               power_w=power_of(s), smi=s["smi"]) for s in sm.samples])
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
