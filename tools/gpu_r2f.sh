cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 300 tools/store_probe > $O/store_probe.txt 2>&1; tail -n 5 $O/store_probe.txt
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "row_chain" 2>&1 | tail -n 5
python -m pytest tests/test_gpu_model.py tests/test_gpu_precision_modes.py tests/test_gpu_next_rows.py -m gpu -q -x 2>&1 | tail -n 8
for i in 1 2; do
EC_TIMELINE=1 python tools/timeline_probe.py > $O/timeline_split_$i.txt 2>&1; tail -n 4 $O/timeline_split_$i.txt
EC_CHAIN_SPLIT=0 EC_TIMELINE=1 python tools/timeline_probe.py > $O/timeline_nosplit_$i.txt 2>&1; tail -n 4 $O/timeline_nosplit_$i.txt
python bench.py --no-cpu-baseline --no-episode --steps 20 > $O/bench_split_$i.json 2>/dev/null; cut -c1-230 $O/bench_split_$i.json
EC_CHAIN_SPLIT=0 python bench.py --no-cpu-baseline --no-episode --steps 20 > $O/bench_nosplit_$i.json 2>/dev/null; cut -c1-230 $O/bench_nosplit_$i.json
done
