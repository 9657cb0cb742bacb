"""In-memory stand-in for the two h5py calls the SkiPose reader makes (h5py is not installed offline).

`File(path, "r")[key][index]` is all that reference lib/dataset/skiPose.py:119-157 and this repo's
lib/dataset/skiPose.py use.  The stand-in serves them from a numpy .npz archive stored under the .h5 name
(tests/golden/assets/ski/ski_test.h5, written by tools/gen_golden.py::write_ski_asset with the dataset keys of
the real asset: 3D [n,51], 2D [n,34], cam_intrinsic [n,3,3], seq, cam, frame).  BOTH readers - the reference's,
imported by tools/gen_golden.py, and this repo's, in tests/test_dataset_files.py - go through this same
module, so the comparison pins the parsing arithmetic, not h5py.
"""
import numpy as np


class _Dataset:
    def __init__(self, arr):
        self._a = arr

    def __len__(self):
        return len(self._a)

    def __getitem__(self, i):
        return np.array(self._a[i])        # h5py hands out a fresh array per read

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype


class File:
    def __init__(self, name, mode="r", **kw):
        if mode != "r":
            raise NotImplementedError("read-only stand-in")
        with np.load(name, allow_pickle=False) as z:
            self._d = {k: z[k] for k in z.files}

    def __getitem__(self, key):
        return _Dataset(self._d[key])

    def keys(self):
        return self._d.keys()

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
