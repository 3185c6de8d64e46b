# round 4, call 17: the 512^3 build soak repeated -- how often does it abort, and with which of this round's switches
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
for rep in 1 2 3 4; do
  for cfg in "new X=1" "old RAMD_TRSV_REFILL=0 RAMD_TRSV_MASKPUB=0"; do
    set -- $cfg; name=$1; shift
    env "$@" timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "soak" > $O/soak_${name}_$rep.log 2>&1
    echo "soak $name $rep rc=$? $(tail -1 $O/soak_${name}_$rep.log | cut -c1-80)"
  done
done
