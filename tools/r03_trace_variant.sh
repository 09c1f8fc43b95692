# kernel trace of one workload on a tuning variant:  bash tools/r03_trace_variant.sh <variant.so> <workload> [packets] [opt=value ...]
LIB=$1; W=$2; N=${3:-1e8}; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/trv
HYP_LIB=$REPO/$LIB timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/trv -o t -- python $REPO/tools/r03_workload.py $W $N "$@" 2>&1 | grep "^$W"
python - <<'P'
import glob, sqlite3
for db in glob.glob('/tmp/trv/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 9"):
        print("%-62s calls %6d total_ms %10.2f avg_us %10.1f  %5.1f%%" % (r[0].split("(")[0].replace("void ", "")[:62], r[1], r[2] / 1e3, r[3], r[4]))
P
