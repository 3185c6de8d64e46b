# round 4, call 18: the 512^3 build soak, 12 runs per setting of the two new switches of the box-tile solve
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04r
mkdir -p $O
cd $R
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  for cfg in "r1m1 RAMD_TRSV_REFILL=1 RAMD_TRSV_MASKPUB=1" "r1m0 RAMD_TRSV_REFILL=1 RAMD_TRSV_MASKPUB=0" "r0m1 RAMD_TRSV_REFILL=0 RAMD_TRSV_MASKPUB=1" "r0m0 RAMD_TRSV_REFILL=0 RAMD_TRSV_MASKPUB=0"; do
    set -- $cfg; name=$1; shift
    env "$@" timeout 300 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "soak" > $O/soak_${name}_$rep.log 2>&1
    echo "soak $name $rep rc=$?"
  done
done 2>&1 | grep -v Aborted | sort | uniq -c | sort -k2
