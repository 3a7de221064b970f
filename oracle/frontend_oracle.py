"""CPU restatement of the reference's tracking front-end loop -- TEST INFRASTRUCTURE ONLY (only tests/ may import it).

Follows dvo_slam/src/local_tracker.cpp:133-216 (initNewLocalMap, update: keyframe + odometry alignment of every frame,
accept callbacks with the all-must-agree combiner of local_tracker.h:46-70, keyframe hand-over) and the pose bookkeeping of
dvo_slam/src/local_map.cpp:170-200, with the two alignments run one after the other through an injected `match`.
Parity unpinned: the reference has no tests for this code.
"""
import numpy as np


def result_is_nan(r):
    return not (np.isfinite(r["T"].sum()) and np.isfinite(r["information"].sum()))


class LocalTracker:
    def __init__(self, match, accept_callbacks):
        """match(ref_image, cur_image, T_init) -> dict(T, information, ...); accept_callbacks: f(r_odometry, r_keyframe) -> bool."""
        self.match, self.accept = match, accept_callbacks
        self.force = False
        self.last_keyframe_pose = np.eye(4)
        self.completed_maps = 0

    def _init_map(self, keyframe, frame, r_odometry, keyframe_pose):
        if result_is_nan(r_odometry):
            r_odometry = dict(r_odometry, T=np.eye(4), information=np.eye(6))
        self.keyframe, self.keyframe_pose = keyframe, keyframe_pose
        self.current = frame
        self.current_pose = keyframe_pose @ r_odometry["T"]          # addKeyframeMeasurement, local_map.cpp:196-200

    def init_new_local_map(self, keyframe, frame, keyframe_pose=None):
        r = self.match(keyframe, frame, np.eye(4))
        self.last_keyframe_pose = r["T"]
        self._init_map(keyframe, frame, r, np.eye(4) if keyframe_pose is None else keyframe_pose)

    def update(self, image):
        r_keyframe = self.match(self.keyframe, image, np.linalg.inv(self.last_keyframe_pose))
        r_odometry = self.match(self.current, image, np.eye(4))
        self.force = self.force or result_is_nan(r_odometry) or result_is_nan(r_keyframe)
        votes = [cb(r_odometry, r_keyframe) for cb in self.accept]   # every slot runs
        if all(votes) and not self.force:
            self.current = image
            self.current_pose = self.keyframe_pose @ r_keyframe["T"]
            self.last_keyframe_pose = r_keyframe["T"]
            switched = False
        else:
            self.force = False
            self.completed_maps += 1
            self._init_map(self.current, image, r_odometry, self.current_pose)
            self.last_keyframe_pose = r_odometry["T"]
            switched = True
        return self.current_pose.copy(), switched
