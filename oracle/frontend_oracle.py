"""CPU restatement of the reference's tracking front-end loop -- TEST INFRASTRUCTURE ONLY (only tests/ may import it).

Follows dvo_slam/src/local_tracker.cpp:133-216 (initNewLocalMap, update: keyframe + odometry alignment of every frame,
accept callbacks with the all-must-agree combiner of local_tracker.h:46-70, keyframe hand-over) and the pose bookkeeping of
dvo_slam/src/local_map.cpp:170-200, with the two alignments run one after the other through an injected `match`.
Pinned: tests/test_oracle_ref.py runs the reference's own keyframe_tracker.cpp + local_tracker.cpp + local_map.cpp (compiled unmodified
into oracle/_ref) on the same frames and demands the same keyframe switches and poses.
"""
import numpy as np


def result_is_nan(r):
    return not (np.isfinite(r["T"].sum()) and np.isfinite(r["information"].sum()))


class LocalTracker:
    def __init__(self, match, accept_callbacks, on_map_initialized=None):
        """match(ref_image, cur_image, T_init) -> dict(T, information, ...); accept_callbacks: f(r_odometry, r_keyframe) -> bool;
        on_map_initialized(r_odometry) is called whenever a local map is opened (local_tracker.cpp:157)."""
        self.match, self.accept, self.on_map_initialized = match, accept_callbacks, on_map_initialized
        self.force = False
        self.last_keyframe_pose = np.eye(4)
        self.completed_maps = 0

    def _init_map(self, keyframe, frame, r_odometry, keyframe_pose):
        if result_is_nan(r_odometry):
            r_odometry = dict(r_odometry, T=np.eye(4), information=np.eye(6))
        self.keyframe, self.keyframe_pose = keyframe, keyframe_pose
        self.current = frame
        self.current_pose = keyframe_pose @ r_odometry["T"]          # addKeyframeMeasurement, local_map.cpp:196-200
        if self.on_map_initialized:
            self.on_map_initialized(r_odometry)

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


class KeyframeSelection:
    """The accept criteria KeyframeTracker installs (dvo_slam/src/keyframe_tracker.cpp:60-72, 86-168), including the in-place
    overwrite of the results on estimate divergence (:125-151).  Results are dicts with T, information, loglik, levels."""

    def __init__(self, max_translational_distance=0.2, min_entropy_ratio=0.91, min_constraint_ratio=0.33):
        self.max_dist, self.min_ratio, self.min_constraints = max_translational_distance, min_entropy_ratio, min_constraint_ratio
        self.last_transform_to_keyframe = np.eye(4)
        self.first = self.sum = self.n = None
        self.trace = []

    @staticmethod
    def value(r):                                      # LogLikelihoodTrackingResultEvaluation, tracking_result_evaluation.cpp:57-60
        return -float(r["loglik"])

    def on_map_initialized(self, r_odometry):          # keyframe_tracker.cpp:86-96
        self.last_transform_to_keyframe = r_odometry["T"].copy()
        self.first = self.sum = self.value(r_odometry)
        self.n = 1.0

    def callbacks(self):
        def evaluation(ro, rk):                        # :105-123
            ratio = self.value(rk) / self.first
            ok = ratio > self.min_ratio
            if ok:
                self.sum += self.value(rk)
                self.n += 1.0
            self.trace.append(ratio)
            return ok

        def divergence(ro, rk):                        # :125-151
            reject = np.linalg.norm(ro["T"][:3, 3]) > 0.1 or np.linalg.norm(rk["T"][:3, 3]) > 1.5 * self.max_dist
            if reject:
                ro["T"] = np.eye(4)
                ro["information"] = np.eye(6) * 0.008 * 0.008
                rk["T"] = self.last_transform_to_keyframe.copy()
            self.last_transform_to_keyframe = rk["T"].copy()
            return not reject

        def distance(ro, rk):                          # :153-156
            return bool(np.linalg.norm(rk["T"][:3, 3]) < self.max_dist)

        def constraint_ratio(ro, rk):                  # :165-168
            level = rk["levels"][-1]
            return level["iterations"][-1]["n"] / level["valid_pixels"] > self.min_constraints
        return [evaluation, divergence, distance, constraint_ratio]
