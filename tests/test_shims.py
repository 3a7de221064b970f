"""The third-party stand-ins behind the drop-in builds (oracle/shim: TBB, g2o), checked on their own: CPU tier.

tests/dropin compiles the reference's pose-graph back end against them twice -- threaded for the engine, serial for the reference's
CPU tracker -- and compares the two graphs; that comparison rests on both walking the same partition of a tbb::parallel_reduce."""
import os
import re
import subprocess

from common import ROOT


def build(flags, name):
    out = os.path.join(ROOT, "tests", "cpp", name)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "oracle", "shim")] + flags +
                          [os.path.join(ROOT, "tests", "cpp", "shim_check.cpp"), "-o", out])
    return out


def test_stand_ins_serial_and_threaded_walk_the_same_partition():
    runs = {}
    for name, flags in (("shim_check_serial", []), ("shim_check_threads", ["-DDVO_SHIM_TBB_THREADS"])):
        text = subprocess.check_output([build(flags, name)], text=True)
        assert text.strip().endswith("shim_check: ok"), text
        runs[name] = [re.sub(r" slots \d+$", "", l) for l in text.splitlines() if l.startswith("n ")]
        slots = [int(l.rsplit(" ", 1)[1]) for l in text.splitlines() if l.startswith("n ")]
        assert max(slots) == (4 if flags else 1)                       # four concurrent branches at most / the caller's own
    assert runs["shim_check_serial"] == runs["shim_check_threads"] and len(runs["shim_check_serial"]) == 15
    assert "n 5 grain 1: [0,1)[1,2)[2,3)[3,5)" not in runs["shim_check_serial"]       # halving, down to single items:
    assert "n 5 grain 1: [0,1)[1,2)[2,3)[3,4)[4,5)" in runs["shim_check_serial"]
