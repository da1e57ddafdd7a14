#!/bin/bash
# round-6 profiles: rocprofv3 kernel stats of the bench command (headline only: --no-other-configs), HBM traffic (PMC, separate passes per
# counter as MI355X_MICROARCH.md prescribes) of the decode kernels at ONE context and of the prefill kernels of one 2048-token prompt
# pass.  Writes gpurun_out/r06prof/{kernel_stats.csv, bench_under_rocprof.json, pmc_traffic.json}; pmc_traffic.json carries its own
# git head / date stamp and the per-class table bench.py reads (newest_pmc_profile).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06prof
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$GRAFT_REPO_ROOT
GIT_HEAD=${GIT_HEAD:-unknown}
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "prof exit $?" >> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -type f ! -name "*stats*" -size +4M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_dec_$c -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 128 1 > /dev/null 2> $OUT/pmc_dec_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_pre_$c -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 4 --reps 2 > /dev/null 2> $OUT/pmc_pre_$c.err
done
python - <<PY
import csv, json, collections, glob, datetime
h, I, g, ctx, V, kvd = 4096, 11008, 128, 2048, 32000, 4096
wb = lambda K, N: K * N / 2 + (K / g) * N * 2.5
alg = {"qkv": wb(h, h) + 2 * wb(h, kvd) + 2 * h + 2 * (h + 2 * kvd), "attn": 2 * (ctx + 1) * kvd * 2 + 2 * (h + 2 * kvd),
       "o_proj": wb(h, h) + 2 * h + 4 * h, "gate_up": 2 * wb(h, I) + 2 * h + 2 * I, "down": wb(I, h) + 2 * I + 4 * h, "head": V * h * 2 + 2 * h + 4 * V}
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/gpu_r05_profiles.sh): decode = build/bench_decoder 8 2048 128 1 "
                 "(context 2048 only), prefill = scripts/prefill_once.py (7B shapes, 4 layers, one 2048-token prompt pass x 2); MI355X, round 6",
       "git_head": "$GIT_HEAD", "collected": datetime.date.today().isoformat(),
       "units": "counters are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "decode": {}, "prefill": {}, "decode_classes": {}}
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
# the per-class table: a decode kernel belongs to the class whose algorithmic bytes its traffic is nearest to (within 20 %)
for n, v in out["decode"].items():
    if v["dispatches"] < 8: continue
    cls = min(alg, key=lambda c: abs(v["hbm_bytes_per_launch"] / alg[c] - 1.0))
    if abs(v["hbm_bytes_per_launch"] / alg[cls] - 1.0) < 0.2 and cls not in out["decode_classes"]:
        out["decode_classes"][cls] = dict(v, kernel=n, algorithmic_bytes=int(alg[cls]), over_algorithmic=round(v["hbm_bytes_per_launch"] / alg[cls], 3))
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["decode_classes"], indent=1)[:2500])
print(json.dumps(out["prefill"], indent=1)[:2500])
PY
find $OUT -name "*.csv" -size +2M -delete
head -16 $OUT/kernel_stats.csv | cut -c1-170
