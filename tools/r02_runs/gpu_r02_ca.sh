#!/bin/bash
# row patterns in the multi-colour sweeps: forced-variant tests, then BiCGStab + MC-SGS at 512^3 with and without
mkdir -p gpurun_out/r02ca
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "variants_forced" > gpurun_out/r02ca/t1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02ca/t1.log
for pat in -1 0; do
RAMD_CSR_PAT=$pat timeout 900 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02ca/b_$pat.json 2> gpurun_out/r02ca/b_$pat.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ca/b_$pat.json').read().strip().splitlines()[-1]); print('pat=$pat bicgstab+mcsgs', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['final_residual'])"
done
