#!/bin/bash
# quick look at the single-mixture lines and timelines (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-quick}
mkdir -p $out
python benchmarks/single_mixture.py 300
rocprofv3 --kernel-trace --output-format csv -d $out/t1 -- python benchmarks/single_mixture.py 60 > /dev/null 2>&1
python benchmarks/single_trace.py $out/t1
python benchmarks/single_mnmf.py 300
rocprofv3 --kernel-trace --output-format csv -d $out/t2 -- python benchmarks/single_mnmf.py 60 > /dev/null 2>&1
python benchmarks/single_trace.py $out/t2
python benchmarks/iva_lines.py
