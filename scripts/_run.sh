mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest.txt 2>&1
EQF_VIO_AMD_LIB=$PWD/build_variants/libeqf_stamps.so python scripts/res_stamps.py 200 1 > gpurun_out/r4e/res_stamps.txt 2>&1
python bench.py --no-cpu-baseline --no-batch64 --no-parity --no-tiled --no-traffic > gpurun_out/r4e/bench_N200.json 2> gpurun_out/r4e/bench_N200.err
python bench.py --filters-per-gpu 8 --steps 880 --warmup 110 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-steady-state > gpurun_out/r4e/bench_b8.json 2>/dev/null
python bench.py --filters-per-gpu 64 --steps 440 --warmup 110 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-steady-state > gpurun_out/r4e/bench_b64.json 2>/dev/null
