cd $GRAFT_REPO_ROOT
O=gpurun_out/h1a; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fp16x1 or row_chain" 2>&1 | tail -n 6
python -m pytest tests/test_gpu_precision_modes.py -m gpu -q -x -s -k "mixed" 2>&1 | grep -v "^$" | tail -n 12
for i in 1 2; do
HEADP=mixed EC_TIMELINE=1 python tools/timeline_probe.py 2>&1 | tail -n 2
HEADP=bf16x3 EC_TIMELINE=1 python tools/timeline_probe.py 2>&1 | tail -n 2
python bench.py --no-cpu-baseline --no-episode --steps 20 --head-precision mixed > $O/bench_mixed_$i.json 2>/dev/null; python tools/bench_line.py < $O/bench_mixed_$i.json | cut -c1-400
python bench.py --no-cpu-baseline --no-episode --steps 20 --head-precision bf16x3 > $O/bench_x3_$i.json 2>/dev/null; python tools/bench_line.py < $O/bench_x3_$i.json | cut -c1-200
done
