# round 4, call 3: masked publication of the box-tile solve (A/B), the multi-colour block form re-expressed as a step list
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_kernels.py tests/test_gpu_shell.py tests/test_gpu_edge_cases.py -m gpu -q -x -k "mcsgs or mcgs or mcilu or MultiColored or multicolo or preconditioner or lusolve or trisolve or ilu or sgs or shell or solvers_vs_golden" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver gmres --precond ilu0 --steps 60 --warmup 10"
for rep in 1 2; do
for v in 0 1; do
  RAMD_TRSV_MASKPUB=$v timeout 600 python $R/bench.py $B > $O/line_mp${v}_$rep.json 2> $O/line_mp${v}_$rep.err; echo "line maskpub=$v $rep rc=$?"
done; done
RAMD_TRSV_MASKPUB=1 timeout 600 python $R/bench.py --matrix shell $B > $O/line_shell_mp1.json 2> $O/line_shell_mp1.err
RAMD_TRSV_MASKPUB=0 timeout 600 python $R/bench.py --matrix shell $B > $O/line_shell_mp0.json 2> $O/line_shell_mp0.err
pmc() { name=$1; ctr=$2; shift; shift
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_${name}_$(echo $ctr | tr ' ' '_' | cut -c1-40) -o bench -- python $R/bench.py $B "$@" > /dev/null 2> $O/pmc_${name}.err
  echo "pmc $ctr $name rc=$?"; }
pmc new FETCH_SIZE --steps 20 --warmup 2
pmc new WRITE_SIZE --steps 20 --warmup 2
python $R/tools/db_summary.py $O
