"""3DPW test-set container (reference lib/dataset/pw3d.py): array contract + `eval_multi`.

`read_data` follows the reference (:177-227): `pw3d_<subset>.npz` with `keypoints3d17_relative`,
`root_cam`, `cam_param{f,c}`, `image_width/height/path`; joints are re-ordered to the H36M layout with
`order` (:76,170-175) and the 2D input is the projection of the 3D label (confidence column = 1).  The
real asset is not available offline; tests/test_dataset_files.py checks the parser bit for bit against the
reference's reader on a small synthetic file of the same format (tests/golden/assets).
"""
import os

import numpy as np

from ._eval import hypothesis_min, subsample

ORDER = [5, 2, 6, 3, 11, 14, 12, 15, 13, 16, 1, 4, 8, 10, 0, 7, 9]


class PW3D:
    def __init__(self, root_path, subset="train", gt2d=True, read_confidence=True, sample_interval=None, rep=1,
                 flip=False, cond_3d_prob=0, abs_coord=False, seq1=False, seq5678=False, rot=False):
        self.root_path, self.subset, self.gt2d, self.abs_coord = root_path, subset, gt2d, abs_coord
        self.sample_interval, self.rep, self.order = sample_interval, rep, ORDER
        self.db_2d, self.db_3d, self.camera_param, self.w, self.h, self.image_name = self.read_data()
        if sample_interval:
            self._sample(sample_interval)
        self.real_data_len = len(self.db_2d)

    @classmethod
    def from_arrays(cls, db_2d, db_3d, camera_param):
        self = object.__new__(cls)
        self.subset, self.rep, self.order = "test", 1, ORDER
        self.db_2d = np.asarray(db_2d, dtype=np.float32)
        self.db_3d = np.asarray(db_3d, dtype=np.float32)
        self.camera_param = np.asarray(camera_param, dtype=np.float32)
        self.real_data_len = len(self.db_2d)
        return self

    def __len__(self):
        return len(self.db_2d) * self.rep

    def _sample(self, k):
        print(f"Class PW3D({self.subset}): sample dataset every {k} frame")
        self.db_2d, self.db_3d, self.camera_param = self.db_2d[::k], self.db_3d[::k], self.camera_param[::k]
        self.w, self.h, self.image_name = self.w[::k], self.h[::k], self.image_name[::k]

    def order_change(self, data):
        out = np.empty_like(data)
        out[self.order] = data
        return out

    def read_data(self):
        path = os.path.join(self.root_path, "pw3d_%s.npz" % self.subset)
        print("loading %s" % os.path.basename(path))
        data = np.load(path, allow_pickle=True)
        kp = data["keypoints3d17_relative"][:, :, :3] + data["root_cam"][:, None, :]
        cam = data["cam_param"].item()
        n = len(kp)
        labels_3d = np.stack([self.order_change(kp[i]) for i in range(n)])
        K = np.zeros((n, 3, 3))
        K[:, 0, 0], K[:, 1, 1] = cam["f"][:, 0], cam["f"][:, 1]
        K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = cam["c"][:, 0], cam["c"][:, 1], 1
        uvw = np.einsum("nij,nkj->nki", K, labels_3d)
        labels_2d = uvw / uvw[:, :, 2:]
        labels_3d = labels_3d.astype(np.float32)
        if not self.abs_coord:
            labels_3d = labels_3d - labels_3d[:, 0:1]
        return (labels_2d.astype(np.float32), labels_3d, K.astype(np.float32),
                np.asarray(data["image_width"], np.float32), np.asarray(data["image_height"], np.float32),
                list(data["image_path"]))

    def gt_centred(self):
        gt = self.db_3d.astype(np.float64)
        return gt - gt[:, 0:1]

    def eval_multi(self, preds, protocol2=False, print_verbose=False, sample_interval=None, valid_ind=None, joint=17, row_offset=0):
        """Best-of-H mean (PA-)MPJPE over poses (reference :286-345)."""
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
