#!/bin/bash
# call 19: per-kernel durations INSIDE the graph replay (kernel trace), previous attention kernel vs this tree, context 4 and 2048
o=gpurun_out/r03s; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
for v in prev new; do
  bin=$GRAFT_REPO_ROOT/build/bench_decoder; [ $v = prev ] && bin=$GRAFT_REPO_ROOT/build/prev/bench_decoder
  rm -rf /tmp/kt_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -- $bin 32 2048 64 2 > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee -a $GRAFT_REPO_ROOT/$o/in_graph_kernel_times.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def cls(n):
    if "dec_attn_kernel" in n: return "attn"
    if "dec_head" in n: return "head"
    if "argmax" in n: return "argmax"
    if "dec_ring_kernel" in n:
        import re
        m = re.search(r"dec_ring_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", n)
        p, e = int(m.group(3)), int(m.group(4))
        return {(1,0):"qkv",(3,1):"o_proj",(0,1):"o_or_down",(1,2):"gate_up"}.get((p,e),"ring")
    return "other"
names = [cls(r["Kernel_Name"]) for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(rows[i]["Start_Timestamp"]) - int(rows[i-1]["End_Timestamp"]) if i else 0 for i in range(len(rows))]
# in-graph instances: an attention kernel whose predecessor is qkv and successor is an o_proj-type kernel
stats = collections.defaultdict(list)
for i in range(1, len(rows) - 1):
    if names[i] == "attn" and names[i-1] == "qkv":
        short = "ILb1E" in rows[i]["Kernel_Name"]
        k = "ctx4" if short else "ctx2048"
        stats[(k, "attn dur")].append(dur[i]); stats[(k, "gap before attn")].append(gap[i]); stats[(k, "gap after attn")].append(gap[i+1])
        stats[(k, "next (o_proj) dur")].append(dur[i+1]); stats[(k, "prev (qkv) dur")].append(dur[i-1])
for k in sorted(stats):
    v = stats[k]; print(sys.argv[2], k, "n", len(v), "mean ns", round(sum(v)/len(v), 1))
PY
done
