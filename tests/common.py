"""Shared helpers of the test-suite: synthetic pairs, golden fixtures, the host emulation of the device code."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")

from oracle import pyoracle as po   # noqa: E402
from dvo_slam_amd import _lib as hl  # noqa: E402

_pairs = {}


def synth(seed, w=640, h=480):
    key = (seed, w, h)
    if key not in _pairs:
        _pairs[key] = po.synth_pair(seed, w, h)
    return _pairs[key]


def oracle_pyramids(pair, levels):
    return po.pyramids_from_pair(pair, levels)


def twist_matrix_error(Ta, Tb):
    """max-abs twist of Ta^-1 Tb"""
    return np.abs(po.se3_log(np.linalg.inv(Ta) @ Tb)).max()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# ---- host emulation of the device headers (tests/emul/emul_device.cpp) ---------------------------------
class EmulLevel(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("ox", C.c_float), ("oy", C.c_float),
                ("R", C.POINTER(C.c_float)), ("A", C.POINTER(C.c_float)), ("B", C.POINTER(C.c_float)), ("n_selected", C.c_int)]


_emul = None


def emul_lib():
    global _emul
    if _emul is None:
        src = os.path.join(HERE, "emul", "emul_device.cpp")
        out = os.path.join(HERE, "emul", "libemul_device.so")
        deps = [src] + [os.path.join(ROOT, "dvo_slam_amd", "csrc", f) for f in
                        ("pixel_math.h", "solver_logic.h", "se3_device.h", "device_types.h", "hd_compat.h", "linear_walk.h")]
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            subprocess.check_call(["g++", "-O2", "-march=native", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unknown-pragmas", "-pthread",
                                   "-shared", "-o", out, src])
        L = C.CDLL(out)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.emul_level_iteration.argtypes = [C.POINTER(EmulLevel), fp, fp, C.c_int, C.POINTER(hl.IterationOut), fp]
        L.emul_match.argtypes = [C.POINTER(EmulLevel), C.POINTER(hl.Config), C.POINTER(hl.Result), C.POINTER(hl.LevelStats), C.c_int,
                                 C.POINTER(hl.IterationStats), C.c_int]
        L.emul_match_speculative.argtypes = L.emul_match.argtypes
        L.emul_exchange_stress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint]
        L.emul_locate_check.argtypes = [C.c_int, C.c_int]
        L.emul_set_schedule.argtypes = [C.c_int]
        L.emul_set_schedule.restype = None
        L.emul_se3_exp.argtypes = [dp, dp]
        L.emul_se3_log.argtypes = [dp, dp]
        L.emul_solve6.argtypes = [dp, dp, dp]
        _emul = L
    return _emul


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class EmulPair:
    """Device-layout planes (R / A / B) built from the oracle's pyramids, as the pyramid kernels would."""

    def __init__(self, ref_pyr, cur_pyr, levels, ithr=0.0, dthr=0.0):
        self.keep = []
        self.arr = (EmulLevel * levels)()
        for l in range(levels):
            rp = [ref_pyr.plane(l, k)[0] for k in range(6)]
            cp = [cur_pyr.plane(l, k)[0] for k in range(6)]
            K = ref_pyr.plane(l, 0)[1]
            n_sel, mask = ref_pyr.select(l, ithr, dthr)
            Zsel = np.where(mask.astype(bool), rp[1], np.float32(np.nan)).astype(np.float32)
            R = np.ascontiguousarray(np.stack([Zsel, rp[0], rp[2], rp[3]], axis=-1), dtype=np.float32)
            A = np.ascontiguousarray(np.stack([cp[0], cp[1], cp[2], cp[3]], axis=-1), dtype=np.float32)
            B = np.ascontiguousarray(np.stack([cp[4], cp[5]], axis=-1), dtype=np.float32)
            self.keep += [R, A, B]
            h, w = rp[0].shape
            self.arr[l] = EmulLevel(w, h, K[0], K[1], K[2], K[3], _fp(R), _fp(A), _fp(B), n_sel)
        self.levels = levels

    def level_iteration(self, level, T34, P_prev=None, first=True):
        T34 = np.ascontiguousarray(np.asarray(T34, dtype=np.float32).reshape(-1)[:12])
        Pp = np.zeros(4, np.float32) if P_prev is None else np.ascontiguousarray(np.asarray(P_prev, np.float32).reshape(-1))
        out = hl.IterationOut()
        L = self.arr[level]
        res = np.empty((L.h, L.w, 2), np.float32)
        rc = emul_lib().emul_level_iteration(C.byref(self.arr[level]), _fp(T34), _fp(Pp), int(first),
                                             C.byref(out), _fp(res))
        return dict(rc=rc, n=out.n, n_selected=out.n_selected, cov=np.array(out.scale_cov), P=np.array(out.precision).reshape(2, 2),
                    neg_ll=out.neg_loglik, A=np.array(out.A).reshape(6, 6), b=np.array(out.b), residuals=res)

    def match(self, cfg, T_init=None, speculative=False, raw=False):
        """cfg: dvo_slam_amd.Config. Returns the same dict layout as pyoracle.match.  speculative: the control flow of the resident
        kernel (emul_match_speculative).  raw: also the raw bytes of result + level + iteration records (for bit comparisons)."""
        ccfg = cfg.to_c()
        res = hl.Result()
        T0 = np.eye(4) if T_init is None else np.asarray(T_init, dtype=np.float64)
        for i, v in enumerate(T0.reshape(-1)):
            res.transformation[i] = v
        nl = cfg.FirstLevel - cfg.LastLevel + 1
        cap = nl * cfg.MaxIterationsPerLevel
        levels = (hl.LevelStats * nl)()
        iters = (hl.IterationStats * cap)()
        fn = emul_lib().emul_match_speculative if speculative else emul_lib().emul_match
        rc = fn(self.arr, C.byref(ccfg), C.byref(res), levels, nl, iters, cap)
        assert rc == 0
        out = result_to_dict(res, levels, iters)
        if raw:
            used = sum(levels[i].n_iterations for i in range(res.n_levels))
            out["raw"] = (bytes(res), bytes(levels)[: res.n_levels * C.sizeof(hl.LevelStats)], bytes(iters)[: used * C.sizeof(hl.IterationStats)])
        return out


def result_to_dict(res, levels, iters):
    out_levels = []
    for li in range(res.n_levels):
        L = levels[li]
        its = []
        for k in range(L.n_iterations):
            s = iters[L.first_iteration_index + k]
            its.append(dict(id=s.id, n=s.valid_constraints, neg_ll=s.tdist_loglik, precision=np.array(s.tdist_precision).reshape(2, 2),
                            prior_ll=s.prior_loglik, x=np.array(s.increment), A=np.array(s.information).reshape(6, 6)))
        out_levels.append(dict(id=L.id, max_valid_pixels=L.max_valid_pixels, valid_pixels=L.valid_pixels, termination=L.termination,
                               iterations=its))
    return dict(T=np.array(res.transformation).reshape(4, 4), information=np.array(res.information).reshape(6, 6),
                loglik=res.loglik, levels=out_levels, entropy=res.entropy, condition_number=res.condition_number,
                constraint_ratio=res.constraint_ratio, constraint_ratio_accepted=res.constraint_ratio_accepted)


def keyframe_statistics_from(result):
    """The reference's host-side derivations from a Result dict (T, information, levels): keyframe_tracker.cpp:165-196,
    tracking_result_evaluation.cpp:52-55, constraint_proposal_voter.cpp:136-140, dense_tracking_config.cpp:138-150."""
    info = result["information"]
    ev = np.sort(np.linalg.eigvalsh(info)) if np.isfinite(info).all() else np.full(6, np.nan)
    level = result["levels"][-1]
    its, term = level["iterations"], level["termination"]
    need = 2 if term in (2, 3) else 1
    accepted = 0.0
    if len(its) >= need:
        accepted = float((its[-2] if term == 2 else its[-1])["n"]) / level["valid_pixels"]
    with np.errstate(all="ignore"):
        return dict(entropy=float(np.log(np.linalg.det(info))), condition_number=float(abs(ev[5] / ev[0])),
                    constraint_ratio=float(its[-1]["n"]) / level["valid_pixels"], constraint_ratio_accepted=accepted)


def tracker_result_to_dict(r):
    """dvo_slam_amd.Result -> the dict layout of pyoracle.match"""
    out_levels = []
    for L in r.Statistics.Levels:
        its = [dict(id=s.Id, n=s.ValidConstraints, neg_ll=s.TDistributionLogLikelihood, precision=s.TDistributionPrecision,
                    prior_ll=s.PriorLogLikelihood, x=s.EstimateIncrement, A=s.EstimateInformation) for s in L.Iterations]
        out_levels.append(dict(id=L.Id, max_valid_pixels=L.MaxValidPixels, valid_pixels=L.ValidPixels, termination=L.TerminationCriterion,
                               iterations=its))
    return dict(T=r.Transformation, information=r.Information, loglik=r.LogLikelihood, levels=out_levels, entropy=r.Entropy,
                condition_number=r.ConditionNumber, constraint_ratio=r.ConstraintRatio, constraint_ratio_accepted=r.ConstraintRatioAccepted)


def oracle_config_from(cfg, mode):
    return po.make_config(cfg.FirstLevel, cfg.LastLevel, cfg.MaxIterationsPerLevel, cfg.Precision, cfg.Mu, cfg.UseInitialEstimate,
                          cfg.IntensityDerivativeThreshold, cfg.DepthDerivativeThreshold, mode)


def compare_runs(a, b):
    """Compare two match() result dicts over the common iteration prefix of every level.

    The iteration COUNT of a level is decided by comparisons at the float noise floor (|x|_inf against
    Precision = 5e-7, -ll against the previous -ll near convergence), so two correct implementations can
    legitimately stop one pass apart; `structure_mismatch` counts such levels and the callers bound it, while
    everything in the common prefix and the final transform must agree."""
    summary = dict(max_x_err=0.0, max_ll_rel=0.0, max_A_rel=0.0, n_mismatch=0, iters=0, structure_mismatch=0, max_iter_count_diff=0)
    assert len(a["levels"]) == len(b["levels"])
    for La, Lb in zip(a["levels"], b["levels"]):
        assert La["id"] == Lb["id"]
        assert La["valid_pixels"] == Lb["valid_pixels"], ("selection size", La["id"], La["valid_pixels"], Lb["valid_pixels"])
        assert La["max_valid_pixels"] == Lb["max_valid_pixels"]
        if len(La["iterations"]) != len(Lb["iterations"]) or La["termination"] != Lb["termination"]:
            summary["structure_mismatch"] += 1
            summary["max_iter_count_diff"] = max(summary["max_iter_count_diff"], abs(len(La["iterations"]) - len(Lb["iterations"])))
        for ia, ib in zip(La["iterations"], Lb["iterations"]):
            summary["iters"] += 1
            summary["n_mismatch"] += int(abs(ia["n"] - ib["n"]) > max(2, 2e-4 * ib["n"]))
            if np.isfinite(ia["neg_ll"]) and np.isfinite(ib["neg_ll"]):
                summary["max_ll_rel"] = max(summary["max_ll_rel"], abs(ia["neg_ll"] - ib["neg_ll"]) / max(1.0, abs(ib["neg_ll"])))
            if np.all(np.isfinite(ia["x"])) and np.all(np.isfinite(ib["x"])):
                summary["max_x_err"] = max(summary["max_x_err"], np.abs(ia["x"] - ib["x"]).max())
                summary["max_A_rel"] = max(summary["max_A_rel"], np.abs(ia["A"] - ib["A"]).max() / np.abs(ib["A"]).max())
    summary["T_err"] = twist_matrix_error(a["T"], b["T"])
    return summary


# ---- the reference's own callers on the engine (tests/dropin) -------------------------------------------------------
_dropin = None


def dropin_api():
    """(library, "dropin_") for the api= argument of pyoracle's ref_* wrappers: tests/dropin/_build/libdvo_dropin.so = the
    reference's unmodified dvo_slam sources + the shared caller code, compiled against include/dvo/ and linked with libdvo_hip.so.
    Built here when the reference tree is present (the built library travels to the GPU box); None when neither is there."""
    global _dropin
    if _dropin is None:
        here = os.path.join(HERE, "dropin")
        import dvo_slam_amd
        dvo_slam_amd.build()
        po.build()
        if os.path.isdir("/root/reference/dvo_slam/src"):
            subprocess.check_call(["make", "-C", here, "-s"])
        path = os.path.join(here, "_build", "libdvo_dropin.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        po.bind_public_api(L, "dropin_")
        L.dropin_engine.restype = C.c_char_p
        L.dropin_level_fields.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                          C.POINTER(C.c_float)]
        _dropin = (L, "dropin_")
    return _dropin
