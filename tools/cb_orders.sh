set -e
cd $GRAFT_REPO_ROOT
for o in none mask morton block6 block9 block12; do
  echo "=== order $o"
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 32,96,256 --order $o
done
for s in 2 4 8; do
  echo "=== stride $s order none / mask"
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 64,128 --stride $s --order none
  timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --channels 64,128 --stride $s --order mask
done
