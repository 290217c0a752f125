for m in 0 1 2 4 8 7 15; do
  echo -n "mask $m: "
  EQF_DEBUG_PROP=$m python scripts/prop_only.py
done
