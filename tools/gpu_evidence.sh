#!/bin/bash
# tools/gpu_evidence.sh -- the one script that runs on the GPU box (through gpurun), replacing the per-experiment gpu_r0X_*.sh of rounds 2-4:
#     gpurun -- 'bash tools/gpu_evidence.sh <round> <section> [<section> ...]'
# Everything lands in gpurun_out/<round>/ (scratch; copy what is to be judged into profiles/<round>/).  Sections:
#   tests            the whole GPU suite                      -> pytest_gpu.txt
#   tests:<expr>     pytest -k <expr>                         -> pytest_<expr>.txt
#   bench            python bench.py                          -> bench_line.json
#   headline         rocprofv3 stats + PMC of the headline launch (tools/profile.sh)         -> gpurun_out/prof_<round>/
#   prof:<cfg>       rocprofv3 stats + PMC of tools/cfg_run.py <cfg> 150 (gl_default | gl_bars | gl_sm | gl_sm64 | configs2 | chain | n1024bars | ring)
#   size:<n>:<s>     the stateless pass at another size (tools/profile.sh --n <n> --streams <s>)
#   i8               the integer pre-smoothing pass alone (tools/bin/rows_i8_bench, built beforehand by tools/rows_i8_bench.sh)
#   overlap[:<streams>]  two / four batches on as many streams against one (tools/sm_overlap.py)
#   single           one GLava instance: the default pipeline per update, batched API and host drop-in (tools/single_instance.py)
#   power            tools/power_probe.py
#   modes:<what>     the stateful chains' speeds (profiles/r06/modes.txt): alloc | counters | multi | stateless | probe (tools/modes*.py, mode_probe.py; every rocprofv3 run bounded)
# GLV_PMC_EXTRA="<counters>" adds a PMC pass to headline / prof / size.
cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-r06}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p "$O"
for sec in "$@"; do
  case $sec in
    tests)      timeout 2400 python -m pytest tests -q -m gpu > "$O/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$O/pytest_gpu.txt" | tail -2 ;;
    tests:*)    k=${sec#tests:}; timeout 1800 python -m pytest tests -q -m gpu -k "$k" > "$O/pytest_$k.txt" 2>&1; tail -5 "$O/pytest_$k.txt" | cut -c1-300 ;;
    bench)      timeout 1500 python bench.py > "$O/bench_line.json" 2> "$O/bench.err"; echo "bench rc $?"; python - "$O/bench_line.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", round(d["value"] / 1e6, 2), "M frames/s", round(d["roofline"]["frac"], 4))
for k, v in d.get("configs", {}).get("gl_default", {}).items():
    if isinstance(v, dict) and "value" in v: print("gl_default." + k, round(v["value"] / 1e6, 2), "M frames/s", round(v["roofline_frac"], 4), v.get("launches_per_step"), v.get("frac_of_28N", ""))
for k in ("configs[2]", "n8192_stateless", "n16384_stateless", "ring_update"):
    v = d.get("configs", {}).get(k)
    if v: print(k, round(v["value"] / 1e6, 2), round(v["roofline_frac"], 4))
v = d.get("configs", {}).get("configs[2]", {}).get("bars_only")
if v: print("configs[2].bars_only", round(v["value"] / 1e6, 2), round(v["roofline_frac"], 4), "live", v.get("live_bins"), "priced at 20N:", round(v.get("frac_of_full_chain", 0), 4))
for k, v in d.get("roofline", {}).get("chains", {}).items(): print("chain", k, round(v["frames_per_s"] / 1e6, 2), round(v["frac"], 4), v.get("placement"))
print("multi_host_threads", d.get("configs", {}).get("multi_host_threads", {}).get("ratio"))
PY
                ;;
    headline)   TRAFFIC_ARGS="--traffic-json $O/hbm_traffic.json --n 4096 --streams 65536 --ops fft --kernel glv_frame_kernel<11,~0,~1,~2,~1,~1,~true,~2,~1,~1,~4,~0,~0>" \
                  bash tools/profile.sh "$R" --no-alt --no-configs --sustained-s 0 > "$O/prof_headline.txt" 2>&1; tail -12 "$O/prof_headline.txt" | cut -c1-200 ;;
    prof:*)     c=${sec#prof:}; bash tools/profile_cmd.sh "${R}_$c" python "$GRAFT_REPO_ROOT/tools/cfg_run.py" "$c" 150 > "$O/prof_$c.txt" 2>&1
                grep -E "glv_|traffic|wall ms" "gpurun_out/prof_${R}_$c/summary.txt" "gpurun_out/prof_${R}_$c/cmd_stats.txt" | cut -c1-260 | head -12 ;;
    size:*)     IFS=: read -r _ n s <<< "$sec"; bash tools/profile.sh "${R}_n$n" --n "$n" --streams "$s" --no-alt --no-configs --sustained-s 0 > "$O/prof_n$n.txt" 2>&1; tail -6 "$O/prof_n$n.txt" | cut -c1-200 ;;
    i8)         for b in tools/bin/rows_i8_bench*; do echo "== $b"; timeout 120 "$b" 4096 32768 20 2>&1 | tail -1; done | tee "$O/i8_bench.txt" ;;
    overlap)    timeout 300 python tools/sm_overlap.py 16384 60 2>&1 | tail -2 | tee "$O/overlap.txt" ;;
    overlap:*)  k=${sec#overlap:}; timeout 300 python tools/sm_overlap.py "$k" 40 2>&1 | tail -5 | tee "$O/overlap_$k.txt" ;;
    single)     timeout 300 python tools/single_instance.py 2>&1 | tail -2 | tee "$O/single_instance.txt" ;;
    modes:alloc)     GLV_MODES_POLICIES="malloc vmm:0 vmm:2 fine" timeout 600 python tools/modes.py alloc chain 6 2>&1 | tee "$O/modes_alloc.txt" ;;
    modes:counters)  timeout 900 python tools/modes.py counters chain 3 2>&1 | cut -c1-400 | tee "$O/modes_counters.txt" ;;
    modes:multi)     (timeout 200 python tools/modes_multi.py chain 6; timeout 200 python tools/modes_multi.py gl 6) 2>&1 | grep -v amdgpu.ids | tee "$O/modes_multi.txt" ;;
    modes:stateless) timeout 200 python tools/modes_stateless.py 6 2>&1 | grep -v amdgpu.ids | tee "$O/modes_stateless.txt" ;;
    modes:probe)     timeout 400 python tools/mode_probe.py chain 240 > "$O/mode_probe_240s.txt" 2>&1; grep "^second" "$O/mode_probe_240s.txt" | awk '{print $3}' | sort -n | sed -n '1p;$p' ;;
    power)      python tools/power_probe.py --seconds 5 > "$O/power.txt" 2>/dev/null; cat "$O/power.txt" ;;
    *)          echo "unknown section $sec" ;;
  esac
done
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > "$O/device.txt"; nproc >> "$O/device.txt"
