#!/bin/bash
# round 4, after the grid rule change: which kernel configuration / workgroup count the autotuner picks per size and class
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python tools/variant_survey.py --out $O/variants.txt --wisdom $O/wisdom_mi355x.txt > /dev/null 2> $O/variants.err
grep -v autotune $O/variants.txt | cut -c1-170 | tail -60
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
