mkdir -p gpurun_out
repo=$PWD
for w in fcc bcc; do
out=$repo/gpurun_out/prof_$w
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 "$repo/tools/time_wide.py" $w 200 > "$out/log.txt" 2>&1
cd $repo
cat $out/log.txt | tail -2
python3 - $out <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:9]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
