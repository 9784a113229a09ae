for b in 0 2 3 4 5 6; do echo "== bpb $b"; SSSPY_AMD_ISS_BPB=$b python benchmarks/iva_lines.py 2>/dev/null | grep "1 mixture"; done
