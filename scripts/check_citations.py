#!/usr/bin/env python
"""Build-container check (needs /root/reference): every `file:line` citation of a reference file in the headers, kernels, oracle,
tests and documents resolves to a file of the reference tree and stays within its length.  tests/test_docs.py runs it when the tree is present."""
import os, re, glob, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); REF = '/root/reference'
ref_files = collections.defaultdict(list)
for d,_,fs in os.walk(REF):
    for f in fs:
        if f.endswith(('.cpp','.h','.hpp','.cfg','.yaml','.txt','.xml')) or f=='Makefile':
            ref_files[f].append(os.path.join(d,f))
own = set()
for d,_,fs in os.walk(ROOT):
    if '.git' in d or 'gpurun_out' in d: continue
    for f in fs: own.add(f)
srcs = glob.glob(ROOT+'/include/**/*.h', recursive=True)+glob.glob(ROOT+'/dvo_slam_amd/csrc/*')+glob.glob(ROOT+'/oracle/*.cpp')+glob.glob(ROOT+'/oracle/*.h')+glob.glob(ROOT+'/oracle/*.py')+glob.glob(ROOT+'/dvo_slam_amd/*.py')+[ROOT+'/DESIGN.md',ROOT+'/HISTORY.md',ROOT+'/INTEGRATION.md',ROOT+'/README.md']+glob.glob(ROOT+'/tests/*.py')
pat = re.compile(r'([A-Za-z0-9_./]+\.(?:cpp|h|hpp|cfg|yaml)):(\d+)(?:-(\d+))?')
bad=[]; n=0
for s in srcs:
    if not os.path.isfile(s): continue
    for ln, line in enumerate(open(s, errors='ignore'), 1):
        for m in pat.finditer(line):
            path, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            base = os.path.basename(path)
            cands = [p for p in ref_files.get(base, []) if p.endswith(path.lstrip('./'))] or (ref_files.get(base, []) if '/' not in path else [])
            if not cands:
                continue        # one of ours, or an abbreviation
            n+=1
            length = max(sum(1 for _ in open(p, errors='ignore')) for p in cands)
            if max(a,b) > length or b < a:
                bad.append("%s:%d cites %s:%s (file has %d lines)" % (os.path.relpath(s,ROOT), ln, path, m.group(0).split(':',1)[1], length))
print(n, "citations checked;", len(bad), "out of range"); print("\n".join(bad[:40]))
