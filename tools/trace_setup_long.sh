cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/setupq2; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o -o t -- python3 tools/setup_timing.py > $o/run.log 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/setupq2/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob("gpurun_out/setupq2/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction","") ))
rows.sort()
# last session creation: find the last k_dense_lauum and show everything within 60 ms before/after with duration > 200 us
t_l=[r for r in rows if "lauum" in r[2]][-1][1]
for s,e,n in rows:
    if t_l-30e6 < s < t_l+40e6 and e-s > 150e3:
        print("%9.2f ms  dur %8.2f ms  %s" % ((s-t_l)/1e6, (e-s)/1e6, n))
PY
rm -f $(find $o -name "*.csv")
