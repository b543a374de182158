python -m pytest tests/test_hip_epilogue.py tests/test_hip_configs.py tests/test_hip_fullsize.py tests/test_binary_search.py -m gpu -x -q --timeout 1500 2>&1 | tail -3
for v in "" "MMW_FACTOR_NO_MFMA=1"; do
  echo "== $v"
  env $v MMW_FACTOR_VERBOSE=1 python bench.py --cpu-iters 0 --steps 20 --warmup 5 2>gpurun_out/q.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['coloring']; print(c['wall_s'], c['Z'], c['rem'], c['per_probe_ms']['factor'])"
  grep "\[factor\] K" gpurun_out/q.err | tail -9 | cut -c1-120
done
