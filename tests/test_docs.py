"""Documentation hygiene (CPU tier): every repository path the top-level documents cite exists."""
import os
import re

from common import ROOT

DOCS = ["DESIGN.md", "HISTORY.md", "INTEGRATION.md", "README.md", os.path.join("profiles", "README.md")]
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


def test_every_option_and_counter_key_is_documented_in_the_header():
    """include/dvo_hip.h names every key dvo_hip_set_option and dvo_hip_get_counter accept (the parser in capi.hip is the list)."""
    src = open(os.path.join(ROOT, "dvo_slam_amd", "csrc", "capi_options.inc")).read()      # (textually included by capi.hip)
    header = open(os.path.join(ROOT, "include", "dvo_hip.h")).read()
    documented = set(re.findall(r'"([a-z_0-9]+)"', header))
    for entry in ("int dvo_hip_set_option(", "int dvo_hip_get_counter("):
        begin = src.index(entry)
        body = src[begin:src.index("\n}\n", begin)]
        keys = set(re.findall(r'std::strcmp\(key, "([a-z_0-9]+)"\) == 0', body))
        assert len(keys) >= 10, entry
        assert not (keys - documented), "%s keys missing from include/dvo_hip.h: %s" % (entry, sorted(keys - documented))
