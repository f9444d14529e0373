#!/bin/bash
# scratch: LDS bank conflicts of the Cholesky's tile kernels (separate --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pmcchol
rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $o -o t -- python3 tools/prof_psd.py lattice > $o/run.log 2>&1
tail -2 $o/run.log
f=$(find $o -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Kernel_Name"]
    key = "syrk" if "k_chol_syrk" in n else "trsm" if "k_chol_trsm" in n else "potrf" if "k_chol_potrf" in n else None
    if key: agg[key][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    c, a = v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-6s LDS bank-conflict cycles %.3e, LDS active cycles %.3e, ratio %.3f" % (k, c, a, c / a if a else 0))
PY
find $o -name "*counter_collection.csv" -delete
