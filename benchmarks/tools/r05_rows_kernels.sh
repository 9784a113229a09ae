#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/hb.py <<'P'
import numpy as np, os, sys
sys.path.insert(0, '.')
from ssspy_amd.linalg import eigh, gmeanmh, sqrtmh
rng = np.random.default_rng(0)
M = 8
x = rng.standard_normal((32800, M, 2*M)) + 1j*rng.standard_normal((32800, M, 2*M))
A = x @ x.swapaxes(-2,-1).conj() / (2*M)
y = rng.standard_normal((32800, M, 2*M)) + 1j*rng.standard_normal((32800, M, 2*M))
B = y @ y.swapaxes(-2,-1).conj() / (2*M)
for _ in range(3):
    eigh(A); gmeanmh(A, B); sqrtmh(A); eigh(A, B)
P
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -- python /tmp/hb.py > /dev/null 2>&1
f=$(find /tmp/prof_h -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    if '_rows' in r['Name']: print("%-40s %5s %10.1f" % (r['Name'].replace('ssspy::','').replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3))
P
LEG_SOURCES=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ipa -- python benchmarks/tools/leg_run.py ilrma_ipa 16 5 > /tmp/leg.txt 2>&1
grep "ms per" /tmp/leg.txt
f=$(find /tmp/prof_ipa -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    print("%-60s %5s %10.1f %6s" % (r['Name'].replace('ssspy::','').replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
timeout 300 python benchmarks/gmnmf_channels.py 8 4 8 2>&1 | grep channels | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ipa or IPA or gauss_mnmf or hermitian or eigh or sqrtmh or gmeanmh or psd" 2>&1 | grep -v "^  \|Warning\|^tests/" | tail -3
