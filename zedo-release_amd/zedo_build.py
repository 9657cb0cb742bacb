"""Build / locate libzedo_hip.so without importing anything that touches a GPU.

Used by zedo_hip/__init__.py (every rank, at import) and by launchers that start several ranks (bench.py --gpus N): the
launcher calls ensure_library() ONCE before it starts the ranks and exports ZEDO_NO_BUILD=1 to them, so that N fresh
processes never run N concurrent `make`s into the same object files.  Ranks started by something else (torchrun) serialise
on an exclusive file lock instead: the first one builds, the others wait and find the finished library.  The Makefile
links to a temporary name and renames, so a library that exists is complete.
"""
import fcntl
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_PATH = os.path.join(PKG, "zedo_hip", "libzedo_hip.so")


class BuildError(ImportError):
    pass


def ensure_library(allow_build=None):
    """-> path of libzedo_hip.so.  Missing library: built with `make -C csrc` under an exclusive lock, unless building is
    switched off (allow_build=False or ZEDO_NO_BUILD=1: a rank of a launcher that has already built) - then, and when
    the build fails, BuildError.  There is no CPU or PyTorch fallback behind this."""
    if os.path.exists(LIB_PATH):
        return LIB_PATH
    if allow_build is None:
        allow_build = os.environ.get("ZEDO_NO_BUILD") != "1"
    if not allow_build:
        raise BuildError(f"{LIB_PATH} is missing and this process may not build it (ZEDO_NO_BUILD=1: the launcher builds "
                         "once, before it starts the ranks).  Build it with __graft_entry__.build() or `make -C "
                         f"{CSRC}`.  The ZeDO hot path has no CPU or PyTorch fallback.")
    try:
        lock = open(os.path.join(CSRC, ".build.lock"), "w")
    except OSError as e:          # a read-only tree cannot be built into either
        raise BuildError(f"{LIB_PATH} is missing and {CSRC} is not writable ({e}).  The ZeDO hot path has no CPU or PyTorch "
                         "fallback.") from e
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):          # nobody built it while this process waited for the lock
                try:
                    subprocess.run(["make", "-C", CSRC, "-j4"], check=True, stdout=subprocess.DEVNULL)
                except Exception as e:  # noqa: BLE001
                    raise BuildError(f"{LIB_PATH} is missing and `make -C {CSRC}` failed ({e}).  Build it with "
                                     "__graft_entry__.build().  The ZeDO hot path has no CPU or PyTorch fallback.") from e
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if not os.path.exists(LIB_PATH):
        raise BuildError(f"`make -C {CSRC}` finished without producing {LIB_PATH}")
    return LIB_PATH
