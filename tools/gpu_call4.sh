set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
for g in 0 256 512 768 1024; do python bench.py --n 8192 --streams 32768 --grid $g --no-alt --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n8192 grid $g', d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"; done > $O/grid8192.txt
for g in 0 256 512 1024; do python bench.py --n 16384 --streams 16384 --grid $g --no-alt --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n16384 grid $g', d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"; done >> $O/grid8192.txt
cat $O/grid8192.txt
cd /tmp && export TMPDIR=/tmp
for lib in r2_n12 r2_n12_noswap; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/lds_$lib -o t -- python $GRAFT_REPO_ROOT/tools/tune.py --streams 32768 --log-modes 1 --reps 2 --iters 3 --lib $GRAFT_REPO_ROOT/glava_amd/csrc/libglvtune_$lib.so > $O/lds_$lib.txt 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/lds_$lib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "glv_frame" in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items(): print("$lib", k, {n: sum(v)/len(v) for n,v in c.items()})
PY
done > $O/lds_ab.txt 2>&1
cat $O/lds_ab.txt
