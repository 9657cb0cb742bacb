"""Human3.6M test-set container (reference lib/dataset/h36m.py): the array contract the sampler needs
(`db_2d [N,17,3]=(u,v,conf)`, `db_3d [N,17,3]` metres, `camera_param [N,3,3]`) and `eval_multi`.

File parsing follows the reference's `read_data` (:206-263): `h36m_<subset>.pkl` is a list of dicts with
`joint_3d_camera` (mm), `joint_3d_image`, `camera_param{fx,fy,cx,cy}`, `action`, `image_path`; detections
come from `h36m_sh_dt_ft.pkl`.  The real assets are Google-Drive downloads that are not available offline;
tests/test_dataset_files.py checks the parser bit for bit against the reference's reader on small synthetic
files of the same format (tests/golden/assets).
"""
import os
import pickle

import numpy as np

from ._eval import hypothesis_min, print_table, subsample


class H36MDataset3D:
    def __init__(self, root_path, subset="train", gt2d=True, read_confidence=True, sample_interval=None, rep=1,
                 flip=False, cond_3d_prob=0, abs_coord=False, rot=False):
        self.root_path, self.subset, self.gt2d = root_path, subset, gt2d
        self.read_confidence, self.sample_interval, self.abs_coord = read_confidence, sample_interval, abs_coord
        self.flip, self.rot, self.rep, self.cond_3d_prob = flip, rot, rep, cond_3d_prob
        self.seq5678 = False
        self.image_name = []
        self.db_2d, self.db_3d, self.gt_dataset, self.camera_param = self.read_data()
        if sample_interval:
            self._sample(sample_interval)
        self.real_data_len = len(self.db_2d)

    @classmethod
    def from_arrays(cls, db_2d, joint_3d_camera_mm, camera_param, actions, abs_coord=True):
        """Build the container from arrays (synthetic data, tests): joint_3d_camera_mm [N,17,3] float64."""
        self = object.__new__(cls)
        self.subset, self.seq5678, self.abs_coord = "test", False, abs_coord
        mm = np.asarray(joint_3d_camera_mm, dtype=np.float64)
        self.gt_dataset = [dict(joint_3d_camera=mm[i], action=int(actions[i])) for i in range(len(mm))]
        lab = mm.astype(np.float32)
        if not abs_coord:
            lab = lab - lab[:, 0:1]
        self.db_3d = lab / 1000.0
        self.db_2d = np.asarray(db_2d, dtype=np.float32)
        self.camera_param = np.asarray(camera_param, dtype=np.float32)
        self.image_name = [""] * len(mm)
        self.real_data_len = len(mm)
        self.rep = 1
        return self

    def __len__(self):
        return len(self.db_2d) * self.rep

    def _sample(self, k):
        print(f"Class H36MDataset({self.subset}): sample dataset every {k} frame")
        self.db_2d, self.db_3d = self.db_2d[::k], self.db_3d[::k]
        self.gt_dataset, self.camera_param = self.gt_dataset[::k], self.camera_param[::k]
        self.image_name = self.image_name[::k]

    def read_data(self):
        path = os.path.join(self.root_path, "h36m_%s.pkl" % self.subset)
        print("loading %s" % os.path.basename(path))
        with open(path, "rb") as f:
            gt_dataset = pickle.load(f)
        n = len(gt_dataset)
        labels_3d = np.empty((n, 17, 3), np.float32)
        labels_img = np.empty((n, 17, 3), np.float32)
        cams = np.zeros((n, 3, 3), np.float32)
        for i, item in enumerate(gt_dataset):
            labels_3d[i] = item["joint_3d_camera"]
            labels_img[i] = item["joint_3d_image"]
            c = item["camera_param"]
            cams[i, 0, 0], cams[i, 1, 1] = np.asarray(c["fx"]).item(), np.asarray(c["fy"]).item()
            cams[i, 0, 2], cams[i, 1, 2], cams[i, 2, 2] = np.asarray(c["cx"]).item(), np.asarray(c["cy"]).item(), 1
            self.image_name.append(item["image_path"])
        if not self.abs_coord:
            labels_3d = labels_3d - labels_3d[:, 0:1]
        labels_3d = labels_3d / 1000.0
        if self.gt2d:
            # dtype as in the reference: float32 pixels, widened to float64 by the appended confidence column
            data_2d = labels_img[..., :2].copy()
            if self.read_confidence:
                data_2d = np.concatenate((data_2d, np.ones((n, 17, 1))), axis=-1)
        else:
            with open(os.path.join(self.root_path, "h36m_sh_dt_ft.pkl"), "rb") as f:
                dt = pickle.load(f)
            data_2d = dt[self.subset]["joint3d_image"][:, :, :2].copy()
            if self.read_confidence:
                data_2d = np.concatenate((data_2d, dt[self.subset]["confidence"].copy()), axis=-1)
            data_2d = data_2d.astype(np.float32)
        return data_2d, labels_3d, gt_dataset, cams

    # ------------------------------------------------------------------ metric
    def gt_centred(self):
        """(gt - gt[0]) / 1000 in float64, as in the reference's inner loop (:400-401)."""
        mm = np.stack([np.asarray(d["joint_3d_camera"], dtype=np.float64) for d in self.gt_dataset])
        return (mm - mm[:, 0:1]) / 1000.0

    def eval_multi(self, preds, protocol2=False, print_verbose=False, sample_interval=None, valid_ind=None, row_offset=0):
        """Best-of-H action-wise MPJPE (reference :365-442).  preds [N, H, 17, 3] (numpy / torch), or
        ("rows", cuda tensor [H*N,17,3]) to keep the sampler output on the device."""
        print("eval multi-hypothesis...")
        gt = self.gt_centred()
        preds, gt, row_offset = subsample(preds, gt, sample_interval, row_offset)
        best, idx = hypothesis_min(preds, gt, protocol2, valid_ind, row_offset)
        k = int(np.argmin(best))
        print(f"maximum MPJPE error: {min(best[k], 1000)} and it is at index: {k}, {idx[k]}")
        actions = np.array([d["action"] for d in self.gt_dataset])
        per_action = [float(np.mean(best[actions == a])) for a in range(2, 17)]
        error = float(np.mean(per_action))
        if print_verbose:
            print_table("p2" if protocol2 else "p1", ["H36M"] + list(range(2, 17)) + ["avg"], per_action + [error])
        self.last_best, self.last_index = best, idx
        return error

    @staticmethod
    def get_skeleton():
        return [[0, 1], [1, 2], [2, 3], [0, 4], [4, 5], [5, 6], [0, 7], [7, 8], [8, 9], [9, 10], [8, 11],
                [11, 12], [12, 13], [8, 14], [14, 15], [15, 16]]
