"""Import-only stand-in so that the reference lib/dataset/skiPose.py can be imported for eval_multi captures (h5py is not installed offline)."""
