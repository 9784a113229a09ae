for b in 0 8 12 16 24 32; do echo "== bpb $b"; SSSPY_AMD_ISS_BPB=$b python benchmarks/iva_lines.py 2>/dev/null | grep ISS; done
