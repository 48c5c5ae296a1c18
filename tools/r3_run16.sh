cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_decode.py tests/test_gpu_e2e.py tests/test_gpu_variant.py -q -x -m gpu 2>&1 | tail -5
bash tools/ab_bench.sh product nosparse product nosparse
python bench.py --probe --batch 256 --no-cpu-baseline 2>&1 | grep -E "phase share|search grid" | head -4
