out=$GRAFT_REPO_ROOT/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python benchmarks/single_mixture.py 300 > $out/single.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/single_trace -- python benchmarks/single_mixture.py 60 > /dev/null 2>&1
python benchmarks/single_trace.py $out/single_trace >> $out/single.txt 2>&1
python benchmarks/single_mnmf.py 300 >> $out/single.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/single_mnmf_trace -- python benchmarks/single_mnmf.py 60 > /dev/null 2>&1
python benchmarks/single_trace.py $out/single_mnmf_trace >> $out/single.txt 2>&1
python benchmarks/iva_lines.py >> $out/single.txt 2>&1
python benchmarks/other_configs.py > $out/other_configs_b1.txt 2>&1
grep -v "^ " $out/single.txt | grep -v amdgpu
grep "FastGaussMNMF" $out/other_configs_b1.txt | cut -c1-200
