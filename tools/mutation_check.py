#!/usr/bin/env python3
"""Sensitivity of the GPU parity suite: nine one-line arithmetic mutations of the HIP path, each of which must turn
at least one `-m gpu` test red (run on the GPU box from the repo root: `python tools/mutation_check.py [out.txt]`).

Each mutant is the product library built with ONE extra -D flag (the hooks are `#ifdef ZEDO_MUT_*` lines in csrc/,
compiled out of the product); the whole `-m gpu` suite runs against it and the failing tests are listed.  The clean
library is rebuilt and re-tested at the end (row "none": must be all green), so no mutant is left in the tree.
"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zedo-release_amd", "csrc")

MUTANTS = [
    ("ZEDO_MUT_GN_EPS", "GroupNorm eps 1e-5 -> 2e-5 (zedo_tile.h epilogue, both math modes; reference model.py:116)"),
    ("ZEDO_MUT_SDE_C", "c_i * (1 + 1e-4): score coefficient of x' = a x + c eps (zedo_capi.hip schedule; sde_lib.py:187-198)"),
    ("ZEDO_MUT_CONF2", "least-squares weight conf^4 -> conf^2 (zedo_geom.hip; simple_zeroshot_opt.py:85-88)"),
    ("ZEDO_MUT_SWITCH", "switch to the least-squares T one iteration late (zedo_capi.hip; run/opt_main.py:203-206)"),
    ("ZEDO_MUT_ARGMIN_TIE", "arg-min ties to the HIGHER hypothesis index (zedo_metric.hip; np.argmin, h36m.py:412)"),
    # round 4: the rewritten IPO kernel and the split post_dense of small batches
    ("ZEDO_MUT_IPO_JOINT", "IPO: the 17th key joint (lane 16 of the half-wave) drops out of the gradient sums (zedo_geom.hip; opt_main.py:189-191)"),
    ("ZEDO_MUT_POST_Q3", "post_dense of batches <= 2048 rows: the fourth K quarter is read from the third (zedo_geom.hip post_reduce_kernel)"),
    # round 5: the split-fp16 mode (every GPU test runs in both modes; these two must turn f16x3 cells red and leave f32 cells green)
    ("ZEDO_MUT_F16_DROP_LH", "f16x3: the W_low x X_high MFMA of every k block is dropped - W as plain fp16 (zedo_gemm16.hip mma)"),
    ("ZEDO_MUT_F16_XLOW0", "f16x3: the low pieces of the activations are lost in the first of the 64 k blocks of every dense layer (zedo_gemm16.hip)"),
]


def build(flag):
    for f in os.listdir(CSRC):
        if f.endswith(".hip"):
            os.utime(os.path.join(CSRC, f))
    subprocess.run(["make", "-C", CSRC, "-j4", f"EXTRA={'-D' + flag if flag else ''}"], check=True,
                   stdout=subprocess.DEVNULL)


def run_suite(modes=None):
    """modes: None = both arithmetic modes of the suite (tests/conftest.py), "f32" / "f16x3" = that one only (ZEDO_TEST_MATH)."""
    t0 = time.time()
    cmd = [sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "--no-header", "-rf", "-p", "no:cacheprovider"]
    if os.environ.get("ZEDO_MUT_DESELECT"):          # e.g. the three-minute end-to-end ensembles: "not lie_inside and not pooled"
        cmd += ["-k", os.environ["ZEDO_MUT_DESELECT"]]
    env = dict(os.environ)
    env.pop("ZEDO_TEST_MATH", None)
    if modes:
        env["ZEDO_TEST_MATH"] = modes
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env)
    out = r.stdout + r.stderr
    failed = sorted(set(re.findall(r"^FAILED (\S+)", out, re.M)))
    m = re.search(r"(\d+) passed", out)
    return failed, int(m.group(1)) if m else 0, time.time() - t0, out


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mutation_check.txt")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    rows, ok = [], True
    try:
        only = os.environ.get("ZEDO_MUT_ONLY")          # e.g. ZEDO_MUT_ONLY=ZEDO_MUT_IPO_JOINT: one mutant (+ the clean rebuild)
        for flag, what in MUTANTS:
            if only and flag != only:
                continue
            build(flag)
            # mutations of code both modes share are hunted in the exact-fp32 cells (half the suite time); the two mutations of the
            # split-fp16 kernels run BOTH modes: they must turn f16x3 cells red and leave every f32 cell green
            f16_only = flag.startswith("ZEDO_MUT_F16_")
            failed, passed, dt, out = run_suite(None if f16_only else "f32")
            if f16_only and any("[f32" in t for t in failed):
                ok = False
            rows.append((flag, what + (" [both modes]" if f16_only else " [f32 cells]"), failed, passed, dt))
            with open(dst + ".partial", "a") as f:        # a run cut short by a time limit still leaves every finished row behind
                f.write(f"-D{flag} | {rows[-1][1]} | {len(failed)} | {passed} | {dt:.0f}\n" + "".join(f"    RED {t}\n" for t in failed))
            ok &= len(failed) > 0
            print(f"{flag}: {len(failed)} red / {passed} green in {dt:.0f} s", flush=True)
    finally:
        build(None)
    failed, passed, dt, out = run_suite()
    rows.append(("none", "the product library (rebuilt without any mutation)", failed, passed, dt))
    ok &= len(failed) == 0
    with open(dst, "w") as f:
        f.write("mutation | what | red tests | green | suite seconds\n")
        for flag, what, failed, passed, dt in rows:
            f.write(f"-D{flag} | {what} | {len(failed)} | {passed} | {dt:.0f}\n")
            for t in failed:
                f.write(f"    RED {t}\n")
        f.write("VERDICT: " + ("every mutation is caught and the product is green\n" if ok else
                               "NOT every mutation is caught (or the product is red)\n"))
    print(open(dst).read())
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
