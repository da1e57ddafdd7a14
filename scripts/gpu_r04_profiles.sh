#!/bin/bash
# round-4 profiles: rocprofv3 kernel stats of the bench command (+ where the slowest launch of each decode class sits in the trace),
# HBM traffic (PMC, separate passes per counter as MI355X_MICROARCH.md prescribes) of the decode kernels at ONE context and of the
# prefill kernels of one 2048-token prompt pass
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04prof
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "prof exit $?" >> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
python - <<PY
import csv, glob, json, collections
# the slowest launch of every kernel: which dispatch it is (index in the trace, how long before it the previous kernel ended)
rows = []
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for i, r in enumerate(rows):
    by[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), i))
out = {}
for name, v in by.items():
    if len(v) < 50: continue
    d = sorted(x[0] for x in v)
    worst, idx = max(v)
    gap = int(rows[idx]["Start_Timestamp"]) - int(rows[idx - 1]["End_Timestamp"]) if idx else None
    out[name] = {"launches": len(v), "median_ns": d[len(d) // 2], "p99_ns": d[int(len(d) * 0.99)], "max_ns": worst, "max_at_dispatch": idx, "of": len(rows),
                 "idle_before_max_ns": gap, "previous_kernel": rows[idx - 1]["Kernel_Name"][:60] if idx else None,
                 "launches_over_3x_median": sum(1 for x in d if x > 3 * d[len(d) // 2])}
json.dump(out, open("$OUT/slowest_launches.json", "w"), indent=1)
PY
find $OUT/prof -type f ! -name "*stats*" -size +4M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_dec_$c -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 128 1 > /dev/null 2> $OUT/pmc_dec_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_pre_$c -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 4 --reps 2 > /dev/null 2> $OUT/pmc_pre_$c.err
done
python - <<PY
import csv, json, collections, glob
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/gpu_r04_profiles.sh): decode = build/bench_decoder 8 2048 128 1 "
                 "(context 2048 only), prefill = scripts/prefill_once.py (7B shapes, 4 layers, one 2048-token prompt pass x 2); MI355X, round 4",
       "units": "counters are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "decode": {}, "prefill": {}}
for part, tag, pat in (("decode", "dec", ("dec_",)), ("prefill", "pre", ("q4_gemm", "flash_prefill", "rms_norm", "column_remap", "tail_reduce"))):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/pmc_%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                if not any(p in n for p in pat) or r["Counter_Name"] != c: continue
                res[n][c].append(float(r["Counter_Value"]))
    for n, v in res.items():
        f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        out[part][n[:110]] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024), "dispatches": len(v["FETCH_SIZE"])}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
find $OUT -name "*.csv" -size +2M -delete
head -16 $OUT/kernel_stats.csv | cut -c1-170
