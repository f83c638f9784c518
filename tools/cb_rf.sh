cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 120 --timeout-method thread -k "plan or order" || exit 1
for s in 1 2 4 8; do
for v in 402 401; do
  echo "=== stride $s variant $v"
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 32,64,96,128,256 --stride $s --order plan --variant $v || exit 1
done
done
