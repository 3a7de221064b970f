"""CPU tier: the C-ABI shared library builds, loads, exports every symbol include/dvo_hip.h declares and
refuses to compute without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import dvo_slam_amd as d
from dvo_slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dvo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvo_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    d.build()
    L = d.lib()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "libdvo_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms, "dvo_slam_amd/_lib.py EXPORTS is out of sync with include/dvo_hip.h"
    assert b"gfx950" in L.dvo_hip_version()


def test_struct_layouts_match_header():
    # sizes the C compiler gives the PODs of include/dvo_hip.h (natural alignment, no packing)
    assert C.sizeof(_lib.Config) == 4 * 4 + 2 * 8 + 2 * 4
    assert C.sizeof(_lib.IterationStats) == 8 + 8 * (1 + 2 + 4 + 1 + 6 + 36)
    assert C.sizeof(_lib.LevelStats) == 24
    assert C.sizeof(_lib.Result) == 8 * (16 + 36 + 1) + 8 + 8 * 4
    assert C.sizeof(_lib.IterationOut) == 8 + 12 + 16 + 4 + 8 * (1 + 36 + 6 + 1)


def test_no_cpu_fallback_without_device():
    L = d.lib()
    if L.dvo_hip_device_count() > 0:
        pytest.skip("a GPU is present; the refusal path is for CPU-only boxes")
    ctx = C.c_void_p()
    assert L.dvo_hip_context_create(0, C.byref(ctx)) == _lib.ERR_NO_DEVICE
    assert not ctx.value
    assert b"no CPU fallback" in L.dvo_hip_last_error(None)
    with pytest.raises(d.DvoHipError):
        d.Context(0)


def test_product_does_not_import_the_oracle():
    """The product path must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "dvo_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "liboracle" not in text and "dvo_oracle.h" not in text, f


def test_config_mirror_defaults():
    c = d.Config()   # dvo_core/src/dense_tracking_config.cpp:27-42
    assert (c.FirstLevel, c.LastLevel, c.MaxIterationsPerLevel, c.Precision, c.Mu, c.UseInitialEstimate) == (3, 1, 100, 5e-7, 0.0, False)
    assert c.getNumLevels() == 4 and c.IsSane() and not c.UseEstimateSmoothing()
    r = d.Result()
    assert r.isNaN()
    r.setIdentity()
    assert not r.isNaN() and r.LogLikelihood == 0.0


def build_facade_example():
    """g++ only: the facade is header-only C++ over the C-ABI, no hipcc needed on the caller's side."""
    import subprocess
    out = os.path.join(ROOT, "tests", "cpp", "facade_example")
    src = os.path.join(ROOT, "tests", "cpp", "facade_example.cpp")
    libdir = os.path.join(ROOT, "dvo_slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
                           "-L" + libdir, "-ldvo_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_cpp_facade_compiles_against_the_reference_api():
    """include/dvo/*.h must keep the reference's call pattern compiling (dvo_benchmark/src/benchmark_slam.cpp:384-415, 486)."""
    d.build()
    exe = build_facade_example()
    assert os.path.exists(exe)


def test_stream_pipeline_object_builds_and_binds_only_the_c_abi():
    """dvo_slam_amd/apps/stream_pipeline.cpp (the benchmark's streaming loop as a C function) is a consumer of include/dvo_hip.h:
    it builds with the host compiler, exports dvo_stream_step and has no undefined symbol outside the C-ABI and libc."""
    import subprocess
    d.build()
    path = os.path.join(os.path.dirname(d.LIB_PATH), "libdvo_stream.so")
    assert os.path.exists(path)
    syms = subprocess.check_output(["nm", "-D", path], text=True).splitlines()
    defined = [l.split()[-1] for l in syms if " T " in l]
    undefined = [l.split()[-1] for l in syms if l.strip().startswith("U ")]
    assert "dvo_stream_step" in defined
    ours = [u for u in undefined if u.startswith("dvo_hip_")]
    assert "dvo_stream_step_host" in defined
    assert sorted(ours) == ["dvo_hip_flush_deferred", "dvo_hip_frames_update_raw_as_ex", "dvo_hip_frames_update_raw_device_as_ex", "dvo_hip_get_counter",
                            "dvo_hip_match_batch"]
    assert all(name in defined for name in ("dvo_stream_lanes_create", "dvo_stream_lanes_submit", "dvo_stream_lanes_collect", "dvo_stream_lanes_destroy"))
    assert all(u.startswith("dvo_hip_") or "GLIBC" in u or u.startswith(("mem", "__")) for u in undefined), undefined


@pytest.mark.gpu
def test_match_batch_over_several_device_contexts_in_one_process(tmp_path):
    """dvo::DenseTracker::matchBatch with the frames spread over 1, 2 and 3 engine contexts (one sub-batch per context on one host
    thread each; on this one-GPU box they are independent contexts of device 0, on an 8-GPU node one per GPU): the same records in
    the caller's order, to the bit."""
    import subprocess
    from dvo_slam_amd import datagen, tum
    d.build()
    seq = datagen.synth_sequence(17, 8, 320, 240)
    tum.write_dataset(str(tmp_path), seq["grey"], seq["depth"], seq["poses"])
    exe = os.path.join(ROOT, "tests", "cpp", "multi_device_check")
    libdir = os.path.join(ROOT, "dvo_slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "multi_device_check.cpp"), "-o", exe, "-L" + libdir, "-ldvo_hip",
                           "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lz"])
    runs = {}
    for contexts in (1, 2, 3):
        out = subprocess.check_output([exe, str(tmp_path / "assoc.txt"), str(contexts)], text=True)
        runs[contexts] = np.array([[float(v) for v in line.split()] for line in out.strip().split("\n")])
        assert runs[contexts].shape == (7, 16)
    # 7 pairs over 2 or 3 contexts give sub-batches of 4+3 / 3+2+2 pairs: with the group size of the resident kernel pinned (the
    # driver does) a pair's record does not depend on its sub-batch, so the records are bit-identical
    assert np.array_equal(runs[1], runs[2]) and np.array_equal(runs[1], runs[3])
    for k in range(7):
        true = np.linalg.inv(seq["poses"][k]) @ seq["poses"][k + 1]
        assert np.abs(runs[1][k].reshape(4, 4) - true).max() < 2e-3


def test_every_option_and_counter_of_the_library_is_documented_in_the_header():
    """dvo_hip_set_option / dvo_hip_get_counter dispatch on strings; include/dvo_hip.h is where an integrator reads what exists."""
    import re
    src = open(os.path.join(ROOT, "dvo_slam_amd", "csrc", "capi_options.inc")).read()      # (textually included by capi.hip)
    header = open(os.path.join(ROOT, "include", "dvo_hip.h")).read()

    def body(signature):
        a = src.index(signature)
        return src[a:src.index("\n}\n", a)]
    options = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', body("int dvo_hip_set_option(dvo_hip_context* ctx, const char* key, int value)")))
    counters = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', body("int dvo_hip_get_counter(dvo_hip_context* ctx, const char* key, long long* value)")))
    assert len(options) >= 10 and len(counters) >= 10, (options, counters)
    missing = sorted(k for k in options | counters if not re.search(r"\b%s\b" % re.escape(k), header))
    assert not missing, "not mentioned in include/dvo_hip.h: %s" % missing


def test_batch_policy_reproduces_the_measured_thresholds(tmp_path):
    """dvo_slam_amd/csrc/batch_policy.h: every batch-size threshold of the host driver as a function of the device's compute units.  On 256
    compute units (MI355X) each rule gives the constant rounds 2-5 measured and pinned; on another device they scale with what the chip
    holds at once.  The rules (pairs per compute unit):
      resident kernel, whole match with results in pinned memory ... 1/16      (16 pairs)
      resident kernel, coarse levels ............................... 1/4       (64)
      resident kernel, first level alone ........................... 7/16      (112); from 1/8 (32) for frames streamed in without taps
      role-aware ingest leaves the taps out ........................ 1/8 frame (32 frames)
      short gathering tiles, level hand-over in the solver steps, 8 log-likelihood workgroups per pair, deferred ingest ... 1 (256)
      16 log-likelihood workgroups per pair ........................ 3/16      (48)
      seven steps ahead of a deferred ingest ....................... 3/4       (192)
      320 x 240 log-likelihood inside the solver step, two-wavefront solver steps ... 2 (512; the latter: beyond)
      background build grid ........................................ one workgroup per compute unit (256)"""
    import subprocess
    src = tmp_path / "policy.cpp"
    src.write_text(r'''
#include <cstdio>
#include "batch_policy.h"
int main(int argc, char** argv) {
  const dvo_hip::BatchPolicy p(argc > 1 ? atoi(argv[1]) : 256);
  int first_no_coarse = 0, first_no_first = 0, first_taps = 0, first_skip = 0, first_short = 0, last_hand = 0, first_fused = 0, first_two = 0, first_lead7 = 0;
  int ll16 = 0, ll8 = 0;
  for (int n = 1; n <= 4096; ++n) {
    if (!first_no_coarse && !p.resident_takes_coarse_levels(n)) first_no_coarse = n;
    if (!first_no_first && !p.resident_first_level_fits(n)) first_no_first = n;
    if (!first_taps && p.taps_missing_prefers_first_level_only(n)) first_taps = n;
    if (!first_skip && p.ingest_skips_taps(n)) first_skip = n;
    if (!first_short && p.short_gather_tiles(n)) first_short = n;
    if (p.level_hand_over(n)) last_hand = n;
    if (!first_fused && p.fused_loglik_on_large_levels(n)) first_fused = n;
    if (!first_two && p.solver_two_waves(n)) first_two = n;
    if (!first_lead7 && p.deferred_ingest_lead(n) == 7) first_lead7 = n;
    if (!ll16 && p.loglik_blocks(n) == 16) ll16 = n;
    if (!ll8 && p.loglik_blocks(n) == 8) ll8 = n;
  }
  std::printf("%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", p.resident_direct_max_pairs(), first_no_coarse, first_no_first, first_taps, first_skip,
              first_short, last_hand, first_fused, first_two, first_lead7, ll16, ll8, p.min_workgroups(8), p.min_workgroups(9), p.min_workgroups(63),
              p.min_workgroups(64), p.defer_ingest_max_pairs(), p.background_build_workgroups());
  return 0;
}
''')
    exe = tmp_path / "policy"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-include", "cstdlib", "-I" + os.path.join(ROOT, "dvo_slam_amd", "csrc"), str(src), "-o", str(exe)])
    at = lambda cus: [int(v) for v in subprocess.check_output([str(exe), str(cus)], text=True).split()]
    #                     direct !coarse !first taps skip short hand fused two  lead7 ll16 ll8 min(8) min(9) min(63) min(64) defer build
    assert at(256) == [16, 65, 113, 32, 32, 256, 256, 512, 513, 192, 48, 256, 512, 1024, 1024, 2048, 256, 256]
    assert at(304) == [19, 77, 134, 38, 38, 304, 304, 608, 609, 228, 57, 304, 608, 608, 1216, 1216, 304, 304]   # (an MI300X: every rule scales with the chip)
