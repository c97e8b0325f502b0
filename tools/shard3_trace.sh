cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for g in 3 1 0; do
  rm -rf /tmp/kb_$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_$g -o run -- python3 $R/tools/kodak_fit.py 3 20000 $g 2>&1 | grep "mode"
  python3 $R/tools/busy_union.py /tmp/kb_$g
  python3 - /tmp/kb_$g <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("  %-62s calls %6s avg %8.2f us total %8.1f ms" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
