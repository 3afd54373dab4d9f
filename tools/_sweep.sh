cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s --output-format csv -- python3 $R/tools/select_bench.py 7680 4320 32 > /dev/null 2>&1
f=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
sel=[r for r in rows if "select_" in r["Kernel_Name"]]
# group sequentially: each topk call = sample, compact, finish
out=collections.defaultdict(list)
for r in sel:
    n=r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    out[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in out.items():
    # 9 k-values x 6 calls each
    per=[sum(v[i*6+1:(i+1)*6])/5 for i in range(len(v)//6)]
    print(n, ["%.0f"%x for x in per])
PY
