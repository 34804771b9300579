cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_ref.py tests/test_f64_order.py -m gpu -x -q 2>&1 | tail -2
timeout 150 python tools/longctx_bench.py 7936 2>&1 | tail -1
timeout 150 python tools/longctx_bench.py 1984 2>&1 | tail -1
