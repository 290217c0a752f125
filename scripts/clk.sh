rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" | head -5
rocm-smi --showperflevel 2>&1 | grep -i perf | head -3
(python bench.py --steps 20000 --warmup 110 --no-cpu-baseline --no-roofline > /tmp/b.log 2>&1 &)
sleep 9; rocm-smi --showclocks 2>&1 | grep -E "sclk" | head -3; rocm-smi --showuse --showpower 2>&1 | grep -E "GPU use|Power" | head -4
sleep 1; rocm-smi --showclocks 2>&1 | grep -E "sclk" | head -3
wait; sleep 4; tail -c 300 /tmp/b.log | head -c 200
