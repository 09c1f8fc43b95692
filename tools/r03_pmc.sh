#!/bin/bash
# rocprofv3 PMC passes of one command (one pass per counter set, kernel trace only alongside); per-kernel sums as JSON + table.
#   usage: SETS="A B;C D" KPAT="walk|lucy" r03_pmc.sh tag <command...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=${SETS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY;FETCH_SIZE;WRITE_SIZE;TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum;GRBM_GUI_ACTIVE"}
IFS=';' read -ra ARR <<< "$SETS"
i=0
for set in "${ARR[@]}"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_$i -o pmc -- "$@" > $OUT/pmc_$i.log 2>&1 || echo "pmc $set failed"
done
python - <<PY | tee $OUT/pmc.txt
import sqlite3, glob, json, re, os
pat = os.environ.get("KPAT", ".")
res = {}
for db in sorted(glob.glob("$OUT/pmc_*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        for cn, kn, v, n in c.execute("select counter_name, kernel_name, sum(value), count(*) from counters_collection group by counter_name, kernel_name"):
            k = kn.split("(")[0].replace("void ", "")
            if re.search(pat, k):
                res.setdefault(k, {})[cn] = v
                res[k]["dispatches"] = n
    except Exception as e:
        print("error", db, e)
json.dump(res, open("$OUT/pmc.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    for cn, v in sorted(d.items()):
        print("    %-28s %.6g" % (cn, v))
PY
grep -h "^mode\|^lucy\|^final\|^{" $OUT/pmc_1.log | tail -3
rm -rf $OUT/pmc_*/
