"""CPU restatement of the reference's loop-closure proposal validation -- TEST INFRASTRUCTURE ONLY (like everything under oracle/: only tests/ may import it).

Follows dvo_slam/src/constraints/constraint_proposal_validator.cpp:69-165 (stage loop, vote collection with early
abort, rejected-proposal removal, keepBest, initial-transformation hand-over), constraint_proposal_voter.cpp:34-186 (the
five voters), constraint_proposal.cpp:30-110 and tracking_result_evaluation.cpp:27-62, one proposal at a time in list
order exactly like the reference; `track` is injected (the oracle's sequential match(), or a table in the logic tests).
Pinned: tests/test_oracle_ref.py::test_proposal_validation_restatement_is_the_references runs the reference's own validator,
voters and evaluation classes (compiled into oracle/_ref) on a synthetic keyframe set and gets the same survivors, order and scores.
"""
import numpy as np

ACCEPT, REJECT = 0, 1
LL_DECREASED, TOO_FEW = 2, 3        # DenseTracker::TerminationCriteria (dense_tracking.h:71-80)


class Keyframe:
    def __init__(self, id, image, pose, evaluation=None):
        self.id, self.image, self.pose, self.evaluation = id, image, np.asarray(pose, float), evaluation


class EntropyEvaluation:
    """EntropyRatioTrackingResultEvaluation (tracking_result_evaluation.cpp:27-50, 52-55)."""

    def __init__(self, first_result):
        self.first = self.value(first_result)
        self.sum, self.n = self.first, 1.0

    @staticmethod
    def value(result):
        return float(np.log(np.linalg.det(result["information"])))

    def add(self, result):
        self.sum += self.value(result)
        self.n += 1.0

    def ratio_with_average(self, result):
        return self.value(result) / self.sum * self.n


class LogLikelihoodEvaluation(EntropyEvaluation):
    """LogLikelihoodTrackingResultEvaluation (tracking_result_evaluation.cpp:57-60): what KeyframeTracker actually
    installs as a keyframe's baseline (keyframe_tracker.cpp:86-96)."""

    @staticmethod
    def value(result):
        return -float(result["loglik"])


class Proposal:
    def __init__(self, reference, current, initial=None):
        self.reference, self.current = reference, current
        self.initial = np.eye(4) if initial is None else np.asarray(initial, float)
        self.result = None
        self.votes = []          # (decision, score)

    @staticmethod
    def with_identity(reference, current):
        return Proposal(reference, current)

    @staticmethod
    def with_relative(reference, current):                       # constraint_proposal.cpp:40-48
        return Proposal(reference, current, np.linalg.inv(current.pose) @ reference.pose)

    def inverse(self):                                           # constraint_proposal.cpp:92-100
        return Proposal(self.current, self.reference, np.linalg.inv(self.initial))

    def total_score(self):
        return sum(s for _, s in self.votes)

    def rejected(self):
        return any(d == REJECT for d, _ in self.votes)

    def accepted(self):
        return not self.rejected()

    def same_frames(self, o):                                    # constraint_proposal.cpp:102-105
        a, b, c, d = self.reference.id, self.current.id, o.reference.id, o.current.id
        return (a == c and b == d) or (a == d and b == c)


def result_is_nan(result):                                       # dense_tracking_config.cpp:110-113
    return not (np.isfinite(result["T"].sum()) and np.isfinite(result["information"].sum()))


# ---- voters: vote(proposal) -> (decision, score); optional create/remove hooks ----------------------------------------
class OdometryConstraintVoter:
    def vote(self, p):
        return (REJECT if abs(p.reference.id - p.current.id) <= 1 else ACCEPT), 0.0


class NaNResultVoter:
    def vote(self, p):
        return (REJECT if result_is_nan(p.result) else ACCEPT), 0.0


class ConstraintRatioVoter:
    def __init__(self, threshold):
        self.threshold = threshold

    def vote(self, p):
        level = p.result["levels"][-1]
        its, term = level["iterations"], level["termination"]
        need = 2 if term in (LL_DECREASED, TOO_FEW) else 1       # dense_tracking_config.cpp:138-143
        ratio = 0.0
        if len(its) >= need:
            last = its[-2] if term == LL_DECREASED else its[-1]   # :145-150
            ratio = float(last["n"]) / float(level["valid_pixels"])
        return (ACCEPT if ratio >= self.threshold else REJECT), 0.0


class TrackingResultEvaluationVoter:
    def __init__(self, threshold):
        self.threshold = threshold

    def vote(self, p):
        ratio = p.reference.evaluation.ratio_with_average(p.result)
        return (ACCEPT if ratio >= self.threshold else REJECT), ratio


class CrossValidationVoter:
    def __init__(self, threshold):
        self.threshold = threshold
        self.pairs = []

    def create(self, proposals):
        for p in list(proposals):
            twin = p.inverse()
            proposals.append(twin)
            self.pairs.append((p, twin))

    def remove(self, proposals):
        for first, second in self.pairs:
            worse = second if (first.total_score() >= second.total_score() and first.accepted()) else first
            for i, q in enumerate(proposals):
                if q is worse:
                    del proposals[i]
                    break
        self.pairs = []

    def vote(self, p):
        twin = next(b if a is p else a for a, b in self.pairs if a is p or b is p)
        diff = twin.result["T"] @ p.result["T"]
        return (ACCEPT if np.linalg.norm(diff[:3, 3]) <= self.threshold else REJECT), 0.0


class Stage:
    def __init__(self, id, cfg, keep_best, voters):
        self.id, self.cfg, self.keep_best, self.voters = id, cfg, keep_best, voters


def keep_best(proposals):                                        # constraint_proposal_validator.cpp:104-130
    i = 0
    while i < len(proposals):
        j = i + 1
        while j < len(proposals):
            if proposals[i].same_frames(proposals[j]):
                if proposals[j].total_score() > proposals[i].total_score():
                    proposals[i], proposals[j] = proposals[j], proposals[i]
                del proposals[j]
            else:
                j += 1
        i += 1


def validate(stages, proposals, track, trace=None):
    """track(stage, proposal) -> result dict(T, information, loglik, levels) for ONE proposal, called in list order."""
    for stage in stages:
        for p in proposals:
            p.votes = []
        for v in stage.voters:
            if hasattr(v, "create"):
                v.create(proposals)
        for p in proposals:
            p.result = track(stage, p)
        for p in proposals:
            for v in stage.voters:
                p.votes.append(v.vote(p))
                if p.votes[-1][0] == REJECT:
                    break
        if trace is not None:
            trace.append([(p.reference.id, p.current.id, list(p.votes)) for p in proposals])
        for v in reversed(stage.voters):
            if hasattr(v, "remove"):
                v.remove(proposals)
        proposals[:] = [p for p in proposals if not p.rejected()]
        if stage.keep_best:
            keep_best(proposals)
        for p in proposals:
            p.initial = np.linalg.inv(p.result["T"])
    return proposals
