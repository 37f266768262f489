for v in "" _rb32; do echo == rows_bench$v; for n in 4096 2048 1024; do tools/bin/rows_bench$v $n 2>&1 | grep -v "amdgpu"; done; done
