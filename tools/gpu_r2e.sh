cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x 2>&1 | tail -n 12
python bench.py > $O/bench.json 2> $O/bench.err; tail -n 3 $O/bench.err; python tools/bench_line.py < $O/bench.json
python tools/refresh_pmc.py --out $O/pmc > $O/pmc.log 2>&1; tail -n 40 $O/pmc.log
