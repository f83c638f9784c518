cd $GRAFT_REPO_ROOT
for s in 1 2 4; do
for v in 402 202 102; do
  echo "=== stride $s variant $v"
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 32,64,96,128,256 --stride $s --order plan --variant $v || exit 1
done
done
