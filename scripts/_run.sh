mkdir -p gpurun_out/r4f
./scripts/micro/rsq_acc > gpurun_out/r4f/rsq.txt 2>&1
for m in 0 1 2; do for wt in 0 1; do
  echo "=== mode $m wt $wt" >> gpurun_out/r4f/f64.txt
  timeout 60 ./scripts/micro/factor64_bench_m$m 0 $wt 4 | head -3 >> gpurun_out/r4f/f64.txt 2>&1
done; done
timeout 60 ./scripts/micro/factor64_bench_m2s 0 0 4 > gpurun_out/r4f/f64_m2_stamps.txt 2>&1
EQF_VIO_AMD_LIB=$PWD/build_variants/libeqf_stamps.so python scripts/res_stamps.py 200 1 > gpurun_out/r4f/res_stamps.txt 2>&1
