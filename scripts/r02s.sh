run() { # name, env..., B
  B=$1; shift
  env "$@" timeout 300 python bench.py --filters-per-gpu $B --steps 440 --warmup 110 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-prewarm > /tmp/b.json 2>/dev/null
  python - "$B" "$*" <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print("B="+sys.argv[1], sys.argv[2], round(d["value"]), d["device_error_flag"], [(r["kernel"], r["launches"], r["avg_us"]) for r in d["kernels"]])
PY
}
for B in 4 8 16 32; do
run $B EQF_PREP_FUSE_MAX=512
run $B EQF_PREP_FUSE_MAX=100000
done
