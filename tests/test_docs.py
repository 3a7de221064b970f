"""Documentation hygiene (CPU tier): every repository path the top-level documents cite exists."""
import os
import re

from common import ROOT

DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md", os.path.join("profiles", "README.md")]
PREFIXES = ("profiles/", "scripts/", "tests/", "dvo_slam_amd/", "oracle/", "include/")


def cited_paths(text):
    for m in re.finditer(r"`([A-Za-z0-9_./\-]+)`", text):
        p = m.group(1)
        if p.startswith(PREFIXES) and not p.endswith("/") and "*" not in p and "<" not in p:
            yield p


def test_cited_repository_paths_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for p in sorted(set(cited_paths(text))):
            path = p.split("::")[0]
            if re.search(r"_build/|/_ref/|gpurun_out|\.so$|/lib/|/bin/", path):      # built artefacts (git-ignored)
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append("%s: %s" % (doc, p))
    assert not missing, "\n".join(missing)


def test_reference_citations_stay_within_the_cited_files():
    """`file:line` citations of reference sources (headers, kernels, oracle, tests, documents) against the reference tree, when it is
    there (the build container; the GPU box has no /root/reference)."""
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference"):
        pytest.skip("no reference tree on this machine")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "check_citations.py")], text=True)
    head = out.splitlines()[0]
    assert int(head.split()[0]) > 300 and "; 0 out of range" in head, out
