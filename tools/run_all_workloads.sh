cd $GRAFT_REPO_ROOT
for w in journal-1pct journal-native er-1pct er-5pct-2k dense-200 er-50k; do
  python bench.py --cpu-iters 0 --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['config'].get('krylov_order'), d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
python bench.py --cpu-iters 0 --expm taylor 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('taylor', d['value'], d['ms_per_step'])"
python tools/batch_throughput.py 8 150
python tools/coloring.py 2>&1 | tail -3
