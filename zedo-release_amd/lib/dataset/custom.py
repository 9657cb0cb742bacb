"""In-the-wild dataset container (reference lib/dataset/custom.py, a template that does not run as shipped:
undefined names at :31,:47,:60 - SURVEY.md 2, row 9).  This version is the working minimum for
run/inference.py: 2D detections + intrinsics, optional 3D labels for `--eval`.

    CustomDataset(db_2d [N,17,3]=(u,v,conf), camera_param [N,3,3], db_3d=None [N,17,3] metres)
    CustomDataset.from_npz(path)   # arrays `db_2d`, `camera_param`, optional `db_3d`
"""
import numpy as np

from ._eval import hypothesis_min


class CustomDataset:
    def __init__(self, db_2d, camera_param, db_3d=None, sample_interval=None):
        self.db_2d = np.asarray(db_2d, dtype=np.float32)
        self.camera_param = np.asarray(camera_param, dtype=np.float32)
        if self.db_2d.ndim != 3 or self.db_2d.shape[1:] != (17, 3) or self.camera_param.shape[1:] != (3, 3):
            raise ValueError("expected db_2d [N,17,3] = (u, v, confidence) and camera_param [N,3,3]")
        self.has_labels = db_3d is not None
        self.db_3d = (np.zeros_like(self.db_2d) if db_3d is None else np.asarray(db_3d, dtype=np.float32))
        if sample_interval:
            self.db_2d, self.db_3d = self.db_2d[::sample_interval], self.db_3d[::sample_interval]
            self.camera_param = self.camera_param[::sample_interval]
        self.real_data_len = len(self.db_2d)

    @classmethod
    def from_npz(cls, path, sample_interval=None):
        d = np.load(path)
        return cls(d["db_2d"], d["camera_param"], d["db_3d"] if "db_3d" in d.files else None, sample_interval)

    def __len__(self):
        return len(self.db_2d)

    def gt_centred(self):
        gt = self.db_3d.astype(np.float64)
        return gt - gt[:, 0:1]

    def eval_multi(self, preds, protocol2=False, print_verbose=False, sample_interval=None, valid_ind=None, joint=17, row_offset=0):
        """Best-of-H mean (PA-)MPJPE (reference :62-108)."""
        if not self.has_labels:
            raise RuntimeError("this dataset has no 3D labels: run without --eval")
        print("eval multi-hypothesis...")
        best, idx = hypothesis_min(preds, self.gt_centred(), protocol2, valid_ind, row_offset)
        error = float(np.mean(best))
        print(f"mean PA-MPJPE : {error}" if protocol2 else f"mean MPJPE : {error}")
        self.last_best, self.last_index = best, idx
        return error
