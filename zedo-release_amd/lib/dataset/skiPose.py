"""SkiPose-PTZ test-set container (reference lib/dataset/skiPose.py): array contract + `eval_multi`.

`read_data` follows the reference (:120-157): `ski_test.h5` with per-frame `3D` [51], `2D` [34] in 0..1,
`cam_intrinsic` [3,3] (both scaled by the 256-pixel crop), `seq`, `cam`, `frame`.  It needs h5py, which is not
installed in the build image: the import happens inside `read_data` and fails loudly there; `from_arrays` and
`eval_multi` do not need it.
"""
import os

import numpy as np

from ._eval import hypothesis_min, subsample


class skiPose:
    def __init__(self, root_path, subset="train", gt2d=True, read_confidence=True, sample_interval=None, rep=1,
                 flip=False, cond_3d_prob=0, abs_coord=False, rot=False):
        self.root_path, self.subset, self.gt2d, self.abs_coord = root_path, subset, gt2d, abs_coord
        self.sample_interval, self.rep = sample_interval, rep
        self.db_2d, self.db_3d, self.camera_param, self.image_name = self.read_data()
        if sample_interval:
            self._sample(sample_interval)
        self.real_data_len = len(self.db_2d)

    @classmethod
    def from_arrays(cls, db_2d, db_3d, camera_param):
        self = object.__new__(cls)
        self.subset, self.rep = "test", 1
        self.db_2d = np.asarray(db_2d, dtype=np.float32)
        self.db_3d = np.asarray(db_3d, dtype=np.float32)
        self.camera_param = np.asarray(camera_param, dtype=np.float32)
        self.real_data_len = len(self.db_2d)
        return self

    def __len__(self):
        return len(self.db_2d) * self.rep

    def _sample(self, k):
        print(f"Class SkiPoseDataset({self.subset}): sample dataset every {k} frame")
        self.db_2d, self.db_3d, self.camera_param = self.db_2d[::k], self.db_3d[::k], self.camera_param[::k]

    def read_data(self):
        import h5py        # not part of the build image; required only for the real asset
        path = os.path.join(self.root_path, "ski_test.h5")
        f = h5py.File(path, "r")
        print("loading %s" % path)
        n = len(f["seq"])
        labels_3d = np.empty((n, 17, 3), np.float32)
        labels_2d = np.ones((n, 17, 3), np.float32)
        cams = np.empty((n, 3, 3), np.float32)
        names = []
        for i in range(n):
            cam = f["cam_intrinsic"][i] * 256
            cam[2, 2] = 1
            cams[i] = cam
            labels_3d[i] = f["3D"][i].reshape(-1, 3)
            labels_2d[i, :, :2] = f["2D"][i].reshape(-1, 2) * 256
            names.append("test/seq_{:03d}/cam_{:02d}/image_{:06d}.png".format(int(f["seq"][i]), int(f["cam"][i]), int(f["frame"][i])))
        if not self.abs_coord:
            labels_3d = labels_3d - labels_3d[:, 0:1]
        return labels_2d, labels_3d, cams, names

    def gt_centred(self):
        gt = self.db_3d.astype(np.float64)
        return gt - gt[:, 0:1]

    def eval_multi(self, preds, protocol2=False, print_verbose=False, sample_interval=None, valid_ind=None, row_offset=0):
        """Best-of-H mean (PA-)MPJPE over poses (reference :159-205)."""
        print("eval multi-hypothesis...")
        preds, gt, row_offset = subsample(preds, self.gt_centred(), sample_interval, row_offset)
        best, idx = hypothesis_min(preds, gt, protocol2, valid_ind, row_offset)
        error = float(np.mean(best))
        print(f"mean PA-MPJPE : {error}" if protocol2 else f"mean MPJPE : {error}")
        self.last_best, self.last_index = best, idx
        return error

    @staticmethod
    def get_skeleton():
        return [[0, 1], [1, 2], [2, 3], [0, 4], [4, 5], [5, 6], [0, 7], [7, 8], [8, 9], [9, 10], [8, 11],
                [11, 12], [12, 13], [8, 14], [14, 15], [15, 16]]
