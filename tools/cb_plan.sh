set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 120 --timeout-method thread -k "plan or order"
for o in mask plan; do
  echo "=== order $o"
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 32,64,96,128,256 --order $o
done
for s in 2 4 8; do
  echo "=== stride $s plan"
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 32,64,128,256 --stride $s --order plan
done
