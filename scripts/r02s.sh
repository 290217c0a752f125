timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
run() { # name, env..., B
  B=$1; shift
  env "$@" timeout 300 python bench.py --filters-per-gpu $B --steps 440 --warmup 110 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-prewarm > /tmp/b.json 2>/dev/null
  python - "$B" "$*" <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print("B="+sys.argv[1], sys.argv[2], round(d["value"]), d["device_error_flag"], [(r["kernel"], r["launches"], r["avg_us"]) for r in d["kernels"]])
PY
}
run 1 X=1
run 1 X=1
for B in 4 8 16 64; do
run $B X=1
run $B EQF_BURST_RING=0
done
for r in 1 0; do
EQF_BURST_RING=$r timeout 600 python bench.py --landmarks 1000 --steps 110 --warmup 22 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-prewarm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('N=1000', round(d['value'],1), [(r['kernel'], r['launches'], r['avg_us']) for r in d['kernels']])"
done
