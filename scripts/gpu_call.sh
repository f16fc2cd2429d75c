#!/bin/bash
# One gpurun call = one named recipe (replaces the per-call wrappers of earlier rounds).  Everything a recipe writes goes
# to gpurun_out/<round>_<name>/ on the GPU box and comes back merged into gpurun_out/ here.
#   gpurun --timeout 1500 -- 'scripts/gpu_call.sh r06 hunt'
set -u
round=$1; name=$2; shift 2
out=gpurun_out/${round}_${name}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
run() { echo "== $*" >> $out/commands.log; ( time "$@" ) ; }
case $name in
  hunt)   # the library-free probe under sharing, then the library-level hunt with the culprit report
    P=scripts/probes/stale_line_probe
    [ -x $P ] || hipcc --offload-arch=gfx950 -O2 -o $P $P.hip
    timeout 60 $P k 3000 > $out/probe_alone_k.txt 2>&1
    for i in 1 2 3 4 5; do timeout 200 $P k 3000 > $out/probe_5x_k_$i.txt 2>&1 & done; wait
    timeout 100 $P idle 40 8 > $out/probe_idle.txt 2>&1 &
    sleep 8
    for i in 1 2 3 4; do timeout 200 $P k 3000 > $out/probe_4x_k_idle_$i.txt 2>&1 & done; wait
    for i in 1 2 3 4 5; do timeout 300 $P h 300 > $out/probe_5x_h_$i.txt 2>&1 & done; wait
    head -3 $out/probe_*.txt | cut -c1-400
    timeout 900 python scripts/stale_read_hunt.py --procs 4 --iters ${HUNT_ITERS:-120} --hold-gb 6 > $out/hunt_parent_6gb.jsonl 2> $out/hunt_parent_6gb.err
    tail -1 $out/hunt_parent_6gb.jsonl | cut -c1-1500
    timeout 900 python scripts/stale_read_hunt.py --procs 4 --iters ${HUNT_ITERS:-120} --hold-gb 0 > $out/hunt_no_parent.jsonl 2> $out/hunt_no_parent.err
    tail -1 $out/hunt_no_parent.jsonl | cut -c1-1500
    ;;
  variants)   # which property of the pass-2 kernel matters: scripts/probes/pass2_variants/build.sh, then the hunt per library
    for lib in ${VARIANTS:-default noslp fminmax plainload vec2 ldsptr}; do
      arg=""; [ $lib != default ] && arg="--lib scratch/pass2_variants/libbm_gar_$lib.so"
      timeout 600 python scripts/stale_read_hunt.py --procs 4 --iters ${HUNT_ITERS:-400} --hold-gb 0 --kinds pass2 $arg > $out/hunt_$lib.jsonl 2> $out/hunt_$lib.err
      tail -1 $out/hunt_$lib.jsonl | cut -c1-600
    done
    ;;
  fixcheck)   # after the packed-fp32 fix: library-free probe, the hunt on the shipped library, bench A/B old flags / new flags, the pair, the suite
    P=scripts/probes/pk_f32_probe
    [ -x $P ] || hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o $P $P.hip
    timeout 120 $P 3000 > $out/pk_alone.txt 2>&1; head -1 $out/pk_alone.txt
    for i in 1 2 3 4 5; do timeout 300 $P 3000 > $out/pk_5x_$i.txt 2>&1 & done; wait
    head -2 $out/pk_5x_*.txt | cut -c1-300
    timeout 900 python scripts/stale_read_hunt.py --procs 4 --iters ${HUNT_ITERS:-400} --hold-gb 0 > $out/hunt_shipped.jsonl 2> $out/hunt_shipped.err
    tail -1 $out/hunt_shipped.jsonl | cut -c1-900
    for side in new old new old; do
      lib=""; [ $side = old ] && lib=scratch/slp/libbm_gar_r5flags.so
      BM_GAR_LIB=$lib timeout 600 python bench.py > $out/bench_${side}_$(date +%s).json 2>> $out/bench.err
    done
    for i in 1 2 3; do
      BM_TEST_POISON=0 timeout 500 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q > $out/pair_$i.log 2>&1
      tail -2 $out/pair_$i.log
    done
    ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
    ;;
  round2)   # new tests of round 6, the 4-experiment test five times, the pk probe with scalar operands, n = 51 timings + PMC, bench twice
    P=scripts/probes/pk_f32_probe
    [ -x $P ] || hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o $P $P.hip
    timeout 120 $P 3000 > $out/pk_alone.txt 2>&1; head -1 $out/pk_alone.txt
    for i in 1 2 3 4 5; do timeout 300 $P 3000 > $out/pk_5x_$i.txt 2>&1 & done; wait
    head -1 $out/pk_5x_*.txt | cut -c1-300
    timeout 900 python -m pytest tests/test_gpu_parity_r6.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -m gpu -x -q > $out/pytest_new.log 2>&1; tail -5 $out/pytest_new.log
    for i in 1 2 3 4 5; do
      timeout 900 python -m pytest tests/test_gpu_reference_loop.py -m gpu -q -k four_experiments > $out/four_$i.log 2>&1; tail -1 $out/four_$i.log
    done
    timeout 300 python scripts/n51_probe.py > $out/n51_probe.txt 2>&1; cat $out/n51_probe.txt
    scripts/pmc_collect.sh $out n51 -- python scripts/n51_probe.py 3 > $out/n51_pmc.txt 2>&1; tail -3 $out/n51_pmc.txt
    for i in 1 2; do timeout 600 python bench.py > $out/bench_$i.json 2> $out/bench_$i.err; wc -c $out/bench_$i.json; done
    ;;
  round3)   # the static window search / Aksel lane pointers: n = 51 again, the whole suite, the bench
    timeout 300 python scripts/n51_probe.py > $out/n51_probe.txt 2>&1; cat $out/n51_probe.txt
    ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
    timeout 600 python bench.py > $out/bench_1.json 2> $out/bench_1.err; wc -c $out/bench_1.json
    ;;
  round4)   # Aksel forms A/B at n = 51, the whole suite, the bench
    for lanes in 0 1 2 0 1 2; do echo "BM_AKSEL_LANES=$lanes"; BM_AKSEL_LANES=$lanes timeout 300 python scripts/n51_probe.py 2>&1 | grep -v amdgpu.ids; done > $out/n51_probe_aksel_forms.txt; grep "LANES\|aksel" $out/n51_probe_aksel_forms.txt
    ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
    timeout 600 python bench.py > $out/bench_1.json 2> $out/bench_1.err; wc -c $out/bench_1.json
    ;;
  round5)   # the sliced Gram reduction: suite, kernel traces of C4 / C3, the per-rank probe, bench twice, then the round-5 failing pair N times
    ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
    for w in bulyan krum; do
      rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -o trace -- python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-traffic > $out/trace_$w.json 2> $out/trace_$w.err
      f=$(ls $out/trace_$w/*/*kernel_stats.csv $out/trace_$w/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${w}_kernel_stats.csv && head -8 $out/${w}_kernel_stats.csv | cut -c1-160
      find $out/trace_$w -name "*kernel_trace.csv" -size +20M -delete
    done
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 scripts/per_rank_probe.py > $out/per_rank_probe.txt 2>&1; tail -12 $out/per_rank_probe.txt
    for i in 1 2; do timeout 600 python bench.py > $out/bench_$i.json 2> $out/bench_$i.err; wc -c $out/bench_$i.json; done
    for i in $(seq 1 ${PAIR_RUNS:-8}); do
      BM_TEST_POISON=0 timeout 500 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q > $out/pair_$i.log 2>&1
      tail -1 $out/pair_$i.log
    done
    ;;
  census)   # every rule of the library under GPU sharing, three launches each on unchanged rows (scripts/stale_read_hunt.py)
    timeout 1500 python scripts/stale_read_hunt.py --procs 4 --iters ${HUNT_ITERS:-250} --hold-gb 0 --kinds ${CENSUS_KINDS:-gram,mean,pass2,median,trmean,phocas,meamed,aksel,cge,brute,krum,bulyan,stats,search} > $out/census.jsonl 2> $out/census.err
    tail -1 $out/census.jsonl | cut -c1-2500
    ;;
  updatestep)   # where a step with the momentum at the update spends its time (kernel trace)
    timeout 300 python scripts/step_update_probe.py > $out/wall.txt 2>&1; cat $out/wall.txt | grep -v amdgpu
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python scripts/step_update_probe.py > $out/traced.txt 2>&1
    f=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -14 $out/kernel_stats.csv | cut -c1-150
    find $out/trace -name "*kernel_trace.csv" -size +20M -delete
    ;;
  slppair)   # the two-build reproducer attempt (scripts/probes/slp_pair_probe): alone, then five processes
    P=scripts/probes/slp_pair_probe/slp_pair_probe
    [ -x $P ] || scripts/probes/slp_pair_probe/build.sh
    timeout 200 $P 3000 > $out/alone.txt 2>&1; head -3 $out/alone.txt | cut -c1-300
    for i in 1 2 3 4 5; do timeout 600 $P ${SLP_ITERS:-6000} > $out/five_$i.txt 2>&1 & done; wait
    head -4 $out/five_*.txt | cut -c1-300
    ;;
  wide)   # 16-byte columns (median / trmean) and 8-byte ones (Aksel) at 29-52 rows: A/B at n = 51, then the parity files
    for w in 1 0 1 0; do echo "BM_COL_WIDE=$w"; BM_COL_WIDE=$w timeout 300 python scripts/n51_probe.py 2>&1 | grep -v amdgpu.ids; done > $out/n51_wide_ab.txt; grep "WIDE\|median\|trmean\|aksel" $out/n51_wide_ab.txt
    timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_gpu_full_size_o1.py -m gpu -x -q > $out/pytest_parity.log 2>&1; tail -3 $out/pytest_parity.log
    ;;
  split)   # Bulyan pass 2 with the sorter inside each window branch: A/B against the library before (scratch/old), then parity
    for lib in new old new old; do
      L=""; [ $lib = old ] && L=scratch/old/libbm_gar_before_split.so
      echo "== $lib"; BM_GAR_LIB=$L timeout 300 python scripts/n51_probe.py 2>&1 | grep "bulyan\|krum"
      BM_GAR_LIB=$L timeout 300 python scripts/bulyan_pass2_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
    done > $out/split_ab.txt 2>&1; cat $out/split_ab.txt | cut -c1-220
    timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_parity_r5.py tests/test_gpu_full_size_o1.py -m gpu -x -q > $out/pytest_parity.log 2>&1; tail -3 $out/pytest_parity.log
    ;;
  devsearch)   # the factor search on the device: parity with the host form, the three forms timed, a kernel trace; Brute's kernel trace
    timeout 900 python -m pytest tests/test_gpu_search_device.py -m gpu -x -q > $out/pytest_search_device.log 2>&1; tail -5 $out/pytest_search_device.log
    timeout 1500 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -m gpu -x -q -k "search or factor" > $out/pytest_search_forms.log 2>&1; tail -3 $out/pytest_search_forms.log
    timeout 600 python scripts/search_forms_probe.py 2>&1 | grep -v amdgpu.ids > $out/search_forms.txt; grep "^rep" $out/search_forms.txt | cut -c1-600
    REPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_search -o search -- python scripts/search_forms_probe.py > $out/prof_search.log 2>&1
    find $out/prof_search -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_brute -o brute -- python scripts/brute_probe.py > $out/prof_brute.log 2>&1
    grep -v amdgpu.ids $out/prof_brute.log | tail -6 | cut -c1-300
    find $out/prof_brute -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
    ;;
  costs)   # library-free: the primitives of a one-workgroup kernel
    P=scripts/probes/one_workgroup_costs
    [ -x $P ] || hipcc --offload-arch=gfx950 -O2 -o $P $P.hip
    timeout 120 $P 4096 > $out/one_workgroup_costs.txt 2>&1; cat $out/one_workgroup_costs.txt
    ;;
  stepsearch)   # a whole step with the search on the device / on the host, two vector lengths; then the whole GPU suite
    for D in 11173962 36546980; do D=$D timeout 600 python scripts/step_search_probe.py 2>&1 | grep -v amdgpu.ids; done > $out/step_search.txt; cat $out/step_search.txt
    ( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
    ;;
  searchv5)   # parity of the device search, then the kernel alone with its phase trace
    timeout 900 python -m pytest tests/test_gpu_search_device.py -m gpu -x -q > $out/pytest_search_device.log 2>&1; tail -5 $out/pytest_search_device.log
    timeout 300 python scripts/search_kernel_probe.py 2>&1 | grep -v amdgpu.ids > $out/search_kernel_probe.txt; cat $out/search_kernel_probe.txt
    timeout 600 python scripts/search_forms_probe.py 2>&1 | grep -v amdgpu.ids > $out/search_forms.txt; grep "^rep" $out/search_forms.txt | cut -c1-600
    ;;
  gcprobe)   # are the bench's 40-130 ms stalls the Python garbage collector?
    timeout 600 python scripts/gc_pause_probe.py 2>&1 | grep -v amdgpu.ids > $out/gc_pause_probe.txt; cat $out/gc_pause_probe.txt | cut -c1-900
    ;;
  stallprobe)   # the one slow search in ten of the bench's first series: where does it wait?
    timeout 600 python scripts/search_stall_probe.py 2>&1 | grep -v amdgpu.ids > $out/search_stall_probe.txt; cat $out/search_stall_probe.txt | cut -c1-700
    timeout 600 python bench.py --no-cpu-baseline --no-traffic > $out/bench_quick.json 2> $out/bench_quick.err; python -c "import json; b=json.load(open('$out/bench_quick.json')); a=b['per_gar']['attack_search_c3_krum']; print('bench without the PMC children and the CPU baselines:', a['scalar_form_each_ms'], a['host_scalar_form_each_ms'])"
    ;;
  stallbench)   # which leg of the default bench run brings the slow search: the CPU baselines or the PMC children?
    for flag in --no-traffic --no-cpu-baseline; do
      timeout 900 python bench.py $flag > $out/bench$flag.json 2> $out/bench$flag.err
      python -c "import json; b=json.load(open('$out/bench$flag.json')); a=b['per_gar']['attack_search_c3_krum']; print('bench $flag:', a['scalar_form_each_ms'], a['host_scalar_form_each_ms'])"
    done
    ;;
  cursor)   # the cursor on the device: its tests, the search tests against the reference loop, the bench's search entries
    timeout 900 python -m pytest tests/test_gpu_search_device.py -m gpu -x -q > $out/pytest_search_device.log 2>&1; tail -5 $out/pytest_search_device.log
    timeout 1500 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -m gpu -x -q -k "search or factor" > $out/pytest_search_forms.log 2>&1; tail -3 $out/pytest_search_forms.log
    timeout 600 python scripts/cursor_forms_probe.py 2>&1 | grep -v amdgpu.ids > $out/cursor_forms.txt; cat $out/cursor_forms.txt | cut -c1-400
    ;;
  cursor2)  # ABI 22 (bm_sqdist2, the median's search as the middle of three): the cursor recipe + the multi-rank file
    timeout 900 python -m pytest tests/test_gpu_search_device.py -m gpu -x -q > $out/pytest_search_device.log 2>&1; tail -5 $out/pytest_search_device.log
    timeout 1500 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -m gpu -x -q -k "search or factor" > $out/pytest_search_forms.log 2>&1; tail -3 $out/pytest_search_forms.log
    timeout 600 python scripts/cursor_forms_probe.py 2>&1 | grep -v amdgpu.ids > $out/cursor_forms.txt; cat $out/cursor_forms.txt | cut -c1-400
    BM_TEST_POISON=0 timeout 600 python -m pytest tests/test_gpu_zz_multirank.py tests/test_gpu_reference_loop.py -m gpu -x -q > $out/pytest_multirank_loop.log 2>&1; tail -3 $out/pytest_multirank_loop.log
    ;;
  bulyaneval)  # ABI 23: Bulyan's second pass, evaluate only — its tests, the search tests, the A/B of the two forms
    timeout 900 python -m pytest tests/test_gpu_search_device.py -m gpu -x -q > $out/pytest_new.log 2>&1; tail -5 $out/pytest_new.log
    timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -m gpu -x -q -k "bulyan" > $out/pytest_bulyan.log 2>&1; tail -3 $out/pytest_bulyan.log
    timeout 600 python scripts/bulyan_search_probe.py 2>&1 | grep -v amdgpu.ids > $out/bulyan_search_probe.txt; cat $out/bulyan_search_probe.txt | cut -c1-400
    ;;
  libab)  # the in-tree library against the kernels of the ABI 22 tree (scratch/old/libbm_gar_abi22_kernels.so: scratch/abi22/build.py), whole default bench lines alternating
    for side in new old new old; do
      lib=""; [ $side = old ] && lib=scratch/old/libbm_gar_abi22_kernels.so
      BM_GAR_LIB=$lib timeout 600 python bench.py ${LIBAB_FLAGS---no-cpu-baseline --no-traffic} > $out/bench_${side}_$(date +%s).json 2>> $out/bench.err
    done
    OUT=$out python - <<'PY' | tee $out/summary.txt
import glob, json, os
for path in sorted(glob.glob(os.environ["OUT"] + "/bench_*.json"), key=lambda p: p.split("_")[-1]):
  b = json.load(open(path)); pg = b["per_gar"]; k = pg["krum_c3"]
  print(path.split("/")[-1][:9], "agg/s %.0f | krum_c3 med %.3f avg %.3f dist %.3f | slab %.3f | m1 %.3f | brute_c3 %.3f | bulyan_c4 %.3f | step_krum %.3f | step_median %.3f | search_krum %.3f" % (
    b["value"], k["median_ms"], k["avg_ms"], k["distance_pass_ms"], pg["krum_c3_slab_rows"]["median_ms"], pg["krum_c3_m1"]["median_ms"],
    pg["brute_c3"]["median_ms"], pg["bulyan_c4_1gpu"]["median_ms"], pg["step_c5_krum"]["median_ms"], pg["step_c5_median"]["median_ms"],
    pg["attack_search_c3_krum"]["scalar_form_ms"]))
PY
    ;;
  searchpmc)  # HBM traffic of the search kernels of ABI 22 / 23: FETCH_SIZE and WRITE_SIZE passes (their own runs, --kernel-trace only)
    timeout 200 python scripts/search_kernels_pmc_probe.py 2>&1 | grep -v amdgpu.ids > $out/algorithmic.txt; cat $out/algorithmic.txt
    for counter in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $counter --kernel-trace --output-format csv -d $out/pmc_$counter -o pmc -- python scripts/search_kernels_pmc_probe.py > $out/pmc_$counter.log 2>&1
    done
    OUT=$out python - <<'PY' | tee $out/traffic.txt
import collections, csv, glob, os
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
  for path in glob.glob(os.environ["OUT"] + f"/pmc_{counter}/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(path)):
      if r["Counter_Name"] == counter:
        per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for disp, v in per.items():
      vals[names[disp]][counter].append(v)
for kern, c in sorted(vals.items()):
  if any(key in kern for key in ("order_pair_kernel<25, 4", "colwise_eval_kernel<3, 0, 4", "sqdist2_kernel<4", "bulyan_pass2_eval_kernel<25, 5, 4")):
    fetch = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1); write = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    print(f"{kern[:70]:70s} launches {len(c['FETCH_SIZE'])}  FETCH_SIZE {fetch:.0f} KiB  WRITE_SIZE {write:.0f} KiB  HBM bytes = 1024 * (2 * FETCH + WRITE) = {1024 * (2 * fetch + write):.0f}")
PY
    find $out -name "*counter_collection.csv" -size +5M -delete
    ;;
  searchtrace)  # kernel trace of the searches against Bulyan and the median (what each launch of a candidate costs)
    for probe in bulyan_search_probe cursor_forms_probe; do
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$probe -o trace -- python scripts/$probe.py > $out/$probe.log 2>&1
      f=$(find $out/trace_$probe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${probe}_kernel_stats.csv && head -14 $out/${probe}_kernel_stats.csv | cut -c1-170
      find $out/trace_$probe -name "*kernel_trace.csv" -size +20M -delete
    done
    ;;
  searchprobe)   # the search kernel alone: warm / cold, 1-64 evaluations
    timeout 300 python scripts/search_kernel_probe.py 2>&1 | grep -v amdgpu.ids > $out/search_kernel_probe.txt; cat $out/search_kernel_probe.txt
    ;;
  pair)   # the failing pair of files as the suite runs them, N times
    for i in $(seq 1 ${PAIR_RUNS:-3}); do
      BM_TEST_POISON=0 timeout 500 python -m pytest tests/test_gpu_full_size_o1.py tests/test_gpu_zz_multirank.py -m gpu -q > $out/pair_$i.log 2>&1
      tail -2 $out/pair_$i.log
    done
    ;;
  final)  # the end of a round: suite with -x, smoke, bench as the driver runs it, a kernel trace of the headline alone, a step with the search
    for D in 11173962 36546980; do D=$D timeout 600 python scripts/step_search_probe.py 2>&1 | grep -v amdgpu.ids; done > $out/step_search.txt; cat $out/step_search.txt
    timeout 600 python scripts/cursor_forms_probe.py 2>&1 | grep -v amdgpu.ids > $out/cursor_forms.txt; cat $out/cursor_forms.txt | cut -c1-400
    ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
    timeout 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
    timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_headline -o trace -- python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 20 > $out/trace_headline.json 2> $out/trace_headline.err
    f=$(find $out/trace_headline -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/headline_kernel_stats.csv && head -6 $out/headline_kernel_stats.csv | cut -c1-200
    find $out/trace_headline -name "*kernel_trace.csv" -size +20M -delete
    ;;
  suite)  # what the driver runs at the end of a round
    ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
    timeout 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
    timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
    ;;
  *)      # anything else: the rest of the command line, logged (an unknown recipe name with nothing behind it is an error)
    [ $# -gt 0 ] || { echo "unknown recipe: $name"; exit 64; }
    "$@" > $out/run.log 2>&1; tail -20 $out/run.log
    ;;
esac
ls -la $out | head -40
