"""Run structure of the association at S2: runs per pose, distinct labels per pose, poses where a landmark's beams
form more than one run (what a hash-free grouping would have to fall back on)."""
import sys
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
name = sys.argv[1] if len(sys.argv) > 1 else "S2"
wl = make_workload(*WORKLOADS[name])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
eng.set_debug(True)
eng.sweep_device("redblack")
lab, _, _ = eng.association()
off = eng.kept_beams()[0]
eng.close()
n = len(off) - 1
head = np.ones(len(lab), dtype=bool)
head[1:] = lab[1:] != lab[:-1]
head[off[:-1][off[:-1] < len(lab)]] = True
pose = np.repeat(np.arange(n), np.diff(off))
runs_lab = lab[head]; runs_pose = pose[head]
runs_per_pose = np.bincount(runs_pose, minlength=n)
# distinct (pose, label) pairs
key = runs_pose.astype(np.int64) * (1 << 32) + (runs_lab.astype(np.int64) + 2)
uniq, cnt = np.unique(key, return_counts=True)
up = (uniq >> 32).astype(np.int64)
ent_per_pose = np.bincount(up, minlength=n)
dup_nonneg = np.zeros(n, dtype=bool)
m = (cnt > 1) & (((uniq & 0xffffffff) - 2) >= 0)
dup_nonneg[up[m]] = True
neg_runs = np.bincount(runs_pose[runs_lab < 0], minlength=n)
print(name, "poses", n, "beams/pose %.1f" % (len(lab) / n), "runs/pose %.2f" % runs_per_pose.mean(), "entries/pose %.2f" % ent_per_pose.mean(),
      "max runs %d" % runs_per_pose.max())
print("poses with a mapped landmark split into several runs: %.2f %%" % (100.0 * dup_nonneg.mean()))
print("gated-out runs per pose: mean %.2f, poses with more than one: %.1f %%" % (neg_runs.mean(), 100.0 * (neg_runs > 1).mean()))
run_len = np.diff(np.flatnonzero(np.r_[head, True]))
print("run length: mean %.2f  p50 %d  p99 %d  max %d" % (run_len.mean(), np.median(run_len), np.percentile(run_len, 99), run_len.max()))
# splits that are NOT the scan's wrap-around (first run and last run of a pose on the same landmark)
first_idx = np.flatnonzero(np.r_[True, runs_pose[1:] != runs_pose[:-1]])
last_idx = np.r_[first_idx[1:] - 1, len(runs_pose) - 1]
wrap = (runs_lab[first_idx] == runs_lab[last_idx]) & (last_idx > first_idx)
wrap_pose = np.zeros(n, dtype=bool); wrap_pose[runs_pose[first_idx[wrap]]] = True
# remove the last run of wrapping poses and recount duplicates
keep = np.ones(len(runs_lab), dtype=bool); keep[last_idx[wrap]] = False
key2 = runs_pose[keep].astype(np.int64) * (1 << 32) + (runs_lab[keep].astype(np.int64) + 2)
u2, c2 = np.unique(key2, return_counts=True)
d2 = np.zeros(n, dtype=bool); d2[(u2[c2 > 1] >> 32).astype(np.int64)] = True
print("wrap-around poses %.2f %%; poses with a split that is not the wrap-around: %.3f %%" % (100.0 * wrap_pose.mean(), 100.0 * d2.mean()))
