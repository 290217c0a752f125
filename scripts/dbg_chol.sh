for m in 0 1 2 4 8 3 7 15; do
  echo -n "mask $m: "
  EQF_DEBUG_CHOL=$m python bench.py --steps 440 --warmup 110 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([ (k['kernel'],k['avg_us']) for k in d['kernels'] if k['kernel']=='k_chol_step'])"
done
