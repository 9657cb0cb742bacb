"""MPI-INF-3DHP test-set container (reference lib/dataset/mpii3dHP.py): array contract, `eval_multi` with
PCK / AUC, and the ground-truth-2D reader.

`read_data` follows the reference's `gt2d=True` branch (:256-312): `mpii3d_<subset>.pkl` is a list of dicts with
`joint_3d_camera` (mm), `joint_2d`, `w`, `h`, `camera_param{fx,fy,cx,cy}`, `imageid`, `valid_i`, `action`
(1-based index into ACTION_CONVERTOR, rewritten in place for the valid test frames).  `_sample` (:236-253)
keeps the valid frames first, then every k-th.  The reference's detection branch (`gt2d=False`, :313-350) does not
run as shipped (it builds `np.array((N, 17, 2))` and indexes it as an [N,17,2] array), so it is not offered.
tests/test_dataset_files.py pins the reader against the reference's on a synthetic file of the same format.
"""
import os
import pickle

import numpy as np

from ._eval import hypothesis_min, print_table, subsample

ACTION_CONVERTOR = [15, 17, 10, 18, 19, 20, 21]
ACTIONS = [15, 10, 17, 18, 19, 20, 21]          # table order of the reference (:501)


class MPII3DHP:
    def __init__(self, root_path, subset="train", gt2d=True, read_confidence=True, sample_interval=None, rep=1,
                 flip=False, cond_3d_prob=0, abs_coord=False, rot=False):
        self.root_path, self.subset, self.gt2d, self.abs_coord = root_path, subset, gt2d, abs_coord
        self.read_confidence, self.sample_interval, self.rep = read_confidence, sample_interval, rep
        self.image_path = []
        self.db_2d, self.db_3d, self.gt_dataset, self.valid_id, self.camera_param = self.read_data()
        if sample_interval:
            self._sample(sample_interval)
        self.real_data_len = len(self.db_2d)

    @classmethod
    def from_arrays(cls, db_2d, joint_3d_camera_mm, camera_param, actions, abs_coord=True):
        self = object.__new__(cls)
        self.subset, self.abs_coord, self.rep = "test", abs_coord, 1
        mm = np.asarray(joint_3d_camera_mm, dtype=np.float64)
        self.gt_dataset = [dict(joint_3d_camera=mm[i], action=int(actions[i])) for i in range(len(mm))]
        lab = mm.astype(np.float32)
        self.db_3d = (lab if abs_coord else lab - lab[:, 0:1]) / 1000.0
        self.db_2d = np.asarray(db_2d, dtype=np.float32)
        self.camera_param = np.asarray(camera_param, dtype=np.float32)
        self.valid_id = np.arange(len(mm))
        self.real_data_len = len(mm)
        return self

    def __len__(self):
        return len(self.db_2d) * self.rep

    def _sample(self, k):
        if len(self.valid_id) != 0:
            v = self.valid_id
            self.db_2d, self.db_3d, self.camera_param = self.db_2d[v, :], self.db_3d[v, :], self.camera_param[v, :]
            self.gt_dataset = [self.gt_dataset[i] for i in v]
            self.image_path = self.image_path[v]
        self.db_2d, self.db_3d, self.camera_param = self.db_2d[::k], self.db_3d[::k], self.camera_param[::k]
        self.gt_dataset, self.image_path = self.gt_dataset[::k], self.image_path[::k]

    def read_data(self):
        if not self.gt2d:
            raise NotImplementedError("MPII3DHP with detected 2D input: the reference's branch (mpii3dHP.py:313-350) "
                                      "does not run as shipped; use --gt")
        path = os.path.join(self.root_path, "mpii3d_%s.pkl" % self.subset)
        print("loading %s" % os.path.basename(path))
        with open(path, "rb") as f:
            gt_dataset = pickle.load(f)
        n = len(gt_dataset)
        labels_3d = np.empty((n, 17, 3), np.float32)
        labels_2d = np.empty((n, 17, 3), np.float32)
        cams = np.zeros((n, 3, 3), np.float32)
        valid = []
        for i, item in enumerate(gt_dataset):
            labels_3d[i], labels_2d[i] = item["joint_3d_camera"], item["joint_2d"]
            c = item["camera_param"]
            cams[i, 0, 0], cams[i, 1, 1], cams[i, 0, 2], cams[i, 1, 2], cams[i, 2, 2] = c["fx"], c["fy"], c["cx"], c["cy"], 1
            self.image_path.append(item["imageid"])
            if self.subset == "test" and int(item["valid_i"]) == 1:
                valid.append(i)
                item["action"] = ACTION_CONVERTOR[int(item["action"]) - 1]
        if not self.abs_coord:
            labels_3d = labels_3d - labels_3d[:, 0:1]
        labels_3d = labels_3d / 1000.0
        data_2d = labels_2d[..., :2].copy()
        if self.read_confidence:
            data_2d = np.concatenate((data_2d, np.ones((n, 17, 1))), axis=-1)     # float64, like the reference
        self.image_path = np.array(self.image_path)
        return data_2d, labels_3d, gt_dataset, np.array(valid), cams

    def gt_centred(self):
        mm = np.stack([np.asarray(d["joint_3d_camera"], dtype=np.float64) for d in self.gt_dataset])
        return (mm - mm[:, 0:1]) / 1000.0

    def eval_multi(self, preds, protocol2=False, print_verbose=False, sample_interval=None, valid_ind=None, row_offset=0):
        """Best-of-H action-wise (PA-)MPJPE plus PCK@150mm / AUC of the selected hypotheses (reference :424-514).
        PCK / AUC need the predictions themselves: pass `preds` as [N,H,17,3] or ("rows", tensor) holding ALL rows."""
        from lib.algorithms.advanced.utils import compute_AUC, compute_PCK
        print("eval multi-hypothesis...")
        preds, gt, row_offset = subsample(preds, self.gt_centred(), sample_interval, row_offset)
        best, idx = hypothesis_min(preds, gt, protocol2, valid_ind, row_offset)
        N = len(best)
        if isinstance(preds, tuple):
            rows = preds[1]
            if row_offset == 0 and rows.shape[0] % N == 0 and (idx >= 0).all():
                import torch
                sel = rows.reshape(-1, N, 17, 3)[torch.as_tensor(idx, device=rows.device, dtype=torch.long),
                                                 torch.arange(N, device=rows.device)].cpu().numpy()
            else:
                sel = None                       # a shard does not hold every selected hypothesis
        else:
            p = preds.cpu().numpy() if hasattr(preds, "cpu") else np.asarray(preds)
            sel = p[np.arange(N), idx]
        if sel is not None:
            gts = self.db_3d - self.db_3d[:, 0:1, :]
            self.last_pck, self.last_auc = compute_PCK(preds=sel, gts=gts), compute_AUC(preds=sel, gts=gts)
            print("PCK :", self.last_pck)
            print("AUC :", self.last_auc)
        actions = np.array([d["action"] for d in self.gt_dataset])
        per_action = [float(np.mean(best[actions == a])) for a in ACTIONS]
        error = float(np.mean(per_action))
        if print_verbose:
            print_table("p2" if protocol2 else "p1", ["3DHP"] + ACTIONS + ["avg"], per_action + [error])
        self.last_best, self.last_index = best, idx
        return error

    @staticmethod
    def get_skeleton():
        return [[0, 1], [1, 2], [2, 3], [0, 4], [4, 5], [5, 6], [0, 7], [7, 8], [8, 9], [9, 10], [8, 11],
                [11, 12], [12, 13], [8, 14], [14, 15], [15, 16]]
