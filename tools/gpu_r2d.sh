cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "row_chain" 2>&1 | tail -n 15
python -m pytest tests/test_gpu_model.py tests/test_gpu_precision_modes.py -m gpu -q -x 2>&1 | tail -n 15
EC_TIMELINE=1 python tools/timeline_probe.py > $O/timeline_chain.txt 2>&1; tail -n 30 $O/timeline_chain.txt
EC_CHAIN=0 EC_TIMELINE=1 python tools/timeline_probe.py > $O/timeline_nochain.txt 2>&1; tail -n 30 $O/timeline_nochain.txt
python bench.py --precision bf16 --no-cpu-baseline --no-episode --steps 20 > $O/bench_chain.json; cut -c1-200 $O/bench_chain.json
EC_CHAIN=0 python bench.py --precision bf16 --no-cpu-baseline --no-episode --steps 20 > $O/bench_nochain.json; cut -c1-200 $O/bench_nochain.json
